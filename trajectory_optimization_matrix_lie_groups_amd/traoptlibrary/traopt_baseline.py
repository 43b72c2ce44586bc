"""Import-only stub: every benchmark_*.py of the reference imports traopt_baseline at module level
(benchmark_SE3_tracking.py:9-10), but the CasADi/IPOPT embedded-Euclidean baselines are outside the
hot path."""


def __getattr__(name):
    def _missing(*a, **k):
        raise NotImplementedError(
            "traopt_baseline.%s is a CasADi/IPOPT comparison baseline of the reference and is not part of "
            "the MI355X hot path" % name)
    return _missing
