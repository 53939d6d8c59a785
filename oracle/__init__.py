"""CPU oracle for the machisplin hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product package ``machisplin_amd`` never does.

PARITY UNPINNED: the reference (jasonleebrown/machisplin) is pure R whose
arithmetic lives in un-vendored, un-pinned CRAN packages (fields, terra, gbm,
randomForest, nnet, earth, kernlab, mgcv); R is absent from the build image and
the reference ships no tests or golden vectors.  The
restatement below is written from the published algorithms of those packages
and anchored on the reference's call sites; it is cross-checked against
independent implementations (scipy RBFInterpolator, scikit-learn evaluators)
in ``tests/``.
"""
