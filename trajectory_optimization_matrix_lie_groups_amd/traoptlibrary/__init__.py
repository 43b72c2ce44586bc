"""Mirror of the reference's ``traoptlibrary`` package for the hot path.

Same module, class and method names, constructor signatures and return containers as
chenghuailin/trajectory_optimization_matrix_lie_groups for

    traopt_utilis      skew / hat / vee / adjoint, matrix <-> quaternion helpers
    traopt_dynamics    BaseDynamics, SE3Dynamics, RigidBodyDynamics, DroneDynamics
    traopt_cost        BaseCost, SE3TrackingQuadraticGaussNewtonCost (+ legacy aliases), ALConstrainedCost
    traopt_constraints BaseConstraint, InputConstraint
    traopt_controller  BaseController, iLQR_Tracking_SE3, iLQR_Tracking_SE3_MS, AL_iLQR_Tracking_SE3_MS
    traopt_baseline    import-only stub (CasADi/IPOPT baselines are out of scope)

Every number these classes return is computed by the HIP extension (C ABI include/tolg.h); the
Python here only marshals arguments and replays histories through the user's callbacks.  New in the
mirror: ``fit_batch`` on the controllers (the joblib fan-out of visualization/perturb_all_compute.py).

``trajectory_optimization_matrix_lie_groups_amd.install_as_traoptlibrary()`` registers the package
under the reference's import name.
"""
