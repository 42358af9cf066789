"""Regenerates the committed golden fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference itself cannot be executed in the build container (jax / cvxpy are not installed), so these
vectors are produced by the repo's CPU oracle (oracle/zopt_oracle.py), which is pinned beforehand by the
reference's own known-answer tests (tests/test_oracle.py + reference_kats.json).  They are regression
anchors for the HIP path ("oracle-generated, KAT-pinned"), NOT outputs of the reference.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import zopt_oracle as zo  # noqa: E402
from tests import problems  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # BASELINE config 2: the first 8 of the 4096 seeded LTI systems (SURVEY 8d "C2"), stored un-tiled.
    A, B, Q, R = problems.random_lti_systems(8, 12, 4, seed=0)
    T = 50
    L = zo.discreteFiniteHorizonLqr(*problems.tile_over_horizon(A, B, Q, R, T), T)
    np.savez_compressed(os.path.join(HERE, "lqr_config2_first8.npz"), A=A, B=B, Q=Q, R=R, T=np.int64(T), L=L)
    print("wrote lqr_config2_first8.npz", L.shape)


if __name__ == "__main__":
    main()
