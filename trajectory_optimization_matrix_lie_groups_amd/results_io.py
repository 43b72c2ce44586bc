"""Result files in the layout the reference's benchmark scripts save and reload
(benchmark_SE3_tracking.py:272-345: `save_results_pickle` / `load_results_pickle`):

    {'prob': {J, dt, q_ref, xi_ref, x0, Q, P, R},
     '<solver>': {'xs': [[q (4,4), xi (6,)], ...], 'us' (N,m), 'J_hist', 'grad_hist'[, 'defect_hist']}, ...}

The reference pickles that dict; here the same nested dict goes to a NumPy `.npz` archive (nothing executable in
the file, loadable with allow_pickle=False): every leaf becomes an array under the key "<section>/<name>", a
state list `xs` the two stacked arrays "<section>/xs_q" and "<section>/xs_xi"."""
import numpy as np


def _put(flat, section, name, value):
    if name in ("xs", "x0") and isinstance(value, (list, tuple)) and len(value) and isinstance(value[0], (list, tuple)):
        flat["%s/%s_q" % (section, name)] = np.stack([np.asarray(x[0], float) for x in value])
        flat["%s/%s_xi" % (section, name)] = np.stack([np.asarray(x[1], float) for x in value])
    elif name == "x0":  # one state [q, xi]
        flat["%s/x0_q" % section] = np.asarray(value[0], float)
        flat["%s/x0_xi" % section] = np.asarray(value[1], float)
    else:
        flat["%s/%s" % (section, name)] = np.asarray(value, dtype=float)


def save_results(filename, data):
    """data: the nested dict above (any number of solver sections)."""
    flat = {}
    for section, entries in data.items():
        for name, value in entries.items():
            _put(flat, section, name, value)
    np.savez_compressed(filename, **flat)
    return filename


def load_results(filename):
    """Inverse of save_results: nested dict, `xs` as a list of [q, xi] pairs, histories as lists of floats."""
    out = {}
    with np.load(filename, allow_pickle=False) as z:
        for key in z.files:
            section, name = key.split("/", 1)
            out.setdefault(section, {})[name] = z[key]
    for section, entries in out.items():
        for base in ("xs", "x0"):
            q, xi = entries.pop(base + "_q", None), entries.pop(base + "_xi", None)
            if q is None:
                continue
            entries[base] = [q, xi] if q.ndim == 2 else [[q[i], xi[i]] for i in range(q.shape[0])]
        for name in list(entries):
            if name.endswith("_hist"):
                entries[name] = [float(v) for v in np.atleast_1d(entries[name])]
            elif name == "dt":
                entries[name] = float(entries[name])
    return out
