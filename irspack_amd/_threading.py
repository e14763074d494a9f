"""Thread-count default, same contract as the reference's ``irspack/_threading.py:5-17``.

The value is validated and carried in the solver config like the reference does;
it does not change how the GPU kernels are launched.
"""
import os
from typing import Optional


def get_n_threads(n_threads: Optional[int]) -> int:
    if n_threads is not None:
        return n_threads
    try:
        cand = os.environ.get("IRSPACK_NUM_THREADS_DEFAULT", os.cpu_count())
        return int(cand or 1)
    except Exception:
        raise ValueError('failed to interpret "IRSPACK_NUM_THREADS_DEFAULT" as an integer.')
