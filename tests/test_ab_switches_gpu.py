"""Every A/B switch of README.md ("Switches for A/B measurements") computes the same results: the switches are read once per
process in static initialisers, so each value gets a fresh child process that runs tests/ab_parity_child.py (small parity
checks of every kernel family against the oracle) with that variable set."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the A/B switches exist only in the -DZM_LAB build of the library (zm_common.h: lab_env); the child processes load that one
LAB_ENVIRON = dict(os.environ, ZOPT_AMD_LIB=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zopt_amd", "csrc",
                                                          "libzopt_amd_lab.so"))

SWITCHES = [{}, {"ZOPT_AMD_LQR_G4": "0"}, {"ZOPT_AMD_LQR_D": "2"}, {"ZOPT_AMD_LQR_PATH": "lds"}, {"ZOPT_AMD_ILQR_PATH": "reg"},
            {"ZOPT_AMD_ROLLOUT_PATH": "generic"}, {"ZOPT_AMD_ILQR_SYNC": "1"}, {"ZOPT_AMD_ILQR_SYNC": "0"},   # 0 is clamped to 1
            {"ZOPT_AMD_ILQR_SYNC": "7"}, {"ZOPT_AMD_MPC_PATH": "lane"}, {"ZOPT_AMD_ILQR_TAIL": "0"}, {"ZOPT_AMD_ILQR_TAIL": "2"}, {"ZOPT_AMD_ROLLOUT_QUAD": "0"}, {"ZOPT_AMD_QUAD_ALL_MAX": "3"}, {"ZOPT_AMD_EXPAND": "group"}, {"ZOPT_AMD_ILQR_SWAP": "0"}, {"ZOPT_AMD_LQR_F32": "tile"}, {"ZOPT_AMD_JAC": "full"}, {"ZOPT_AMD_HES": "dense"}]


@pytest.mark.parametrize("env", SWITCHES, ids=[",".join(f"{k}={v}" for k, v in e.items()) or "defaults" for e in SWITCHES])
def test_switch_value_keeps_parity(env):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ab_parity_child.py")], env=dict(LAB_ENVIRON, **env),
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "AB-PARITY-OK" in out.stdout, (out.stdout[-500:], out.stderr[-1500:])
