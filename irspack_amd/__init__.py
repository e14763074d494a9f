"""irspack_amd — MI355X (gfx950) native implementation of irspack's compiled hot path.

The package mirrors the module paths the reference's Python imports
(``irspack.recommenders._ials_core``, ``._knn``,
``irspack.evaluation._core_evaluator``) on top of ``libirspack_amd.so`` (hand-written
HIP kernels behind the C ABI in ``include/irspack_amd.h``).  There is no CPU
fallback: compute entry points raise when no HIP device is visible.
"""

__version__ = "0.1.0"
