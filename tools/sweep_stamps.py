#!/usr/bin/env python3
"""Phase shares of the ring sweep kernel from its -DZM_SWEEP_LAB build (s_memtime stamps of wave 0 of block 0, summed over a solve).
build:  make -C zopt_amd/csrc EXTRA=-DZM_SWEEP_LAB OBJDIR=_obj_sl OUT=libzopt_amd_sl.so
usage:  ZOPT_AMD_LIB=zopt_amd/csrc/libzopt_amd_sl.so python tools/sweep_stamps.py [--ddp] [--batch 1024]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tools import secondary_bench as sb
from zopt_amd import _lib, ilqrUtils

ap = argparse.ArgumentParser()
ap.add_argument("--ddp", action="store_true")
ap.add_argument("--batch", type=int, default=1024)
args = ap.parse_args()
model, cost, x0, ug = sb.config3_problem(ddp=args.ddp) if "ddp" in sb.config3_problem.__code__.co_varnames else sb.config3_problem()
x0, ug = x0[: args.batch], ug[: args.batch]
solver = ilqrUtils.differentialDynamicProgramming if args.ddp else ilqrUtils.iterativeLqr
lib = _lib.lib()
out = (ctypes.c_ulonglong * 10)()
solver(model, cost, cost, x0, ug)
lib.zm_lab_sweep_stamps(out, 1)
solver(model, cost, cost, x0, ug)
lib.zm_lab_sweep_stamps(out, 1)
v = np.array(list(out), dtype=np.float64)
names = ["wait for DMA", "operand reads (+ contraction), DMA issue", "projection / G MFMAs / vector terms", "LDS exchange", "4x4 solve",
         "store, value terms, V' MFMAs, v_x"]
tot = v[:6].sum()
launches = v[8]
print(f"launches {int(launches)}, ticks per launch {tot / max(launches, 1):.0f} (s_memtime ticks: 100 MHz)")
for n_, t in zip(names, v[:6]):
    print(f"  {n_:48s} {t / tot:6.1%}")
