"""The schedule switches of the persistent factorisation (chol_mega.hip) are read once per process from the environment: every variant
that can be selected -- the round-3 schedule, the edge regime at the head, halves only for a tile's last windows, chain tiles' queues,
odd window / slack settings -- must factor correctly.  One child process per variant (tools/potrf_time.py with its residual check:
|L L' - A| / |A| on a device-resident s.p.d. matrix), sizes with 16 and 50 block columns (the latter has the three-streamed-row middle,
so the edge regime, the reserve workgroups and the tail's halves are all in play)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = [
    {},                                                                       # the defaults
    {"MRBF_MEGA_TAIL": "0", "MRBF_MEGA_TAILHALF": "0"},                       # the round-3 schedule
    {"MRBF_MEGA_HEAD": "6", "MRBF_MEGA_TAIL": "12", "MRBF_MEGA_RESERVE": "20"},
    {"MRBF_MEGA_TAILHALF": "30", "MRBF_MEGA_TAILHALF_W": "2"},
    {"MRBF_MEGA_CHAINQ": "1", "MRBF_MEGA_CBOOST": "8"},
    {"MRBF_MEGA_CHAINQ": "1", "MRBF_MEGA_HALF_COLS": "3", "MRBF_MEGA_WIN": "3", "MRBF_MEGA_SLACK": "2", "MRBF_MEGA_SLACK_CHAIN": "4"},
    {"MRBF_MEGA_SROWS": "2", "MRBF_MEGA_PSTREAM": "1", "MRBF_MEGA_CHAIN": "12", "MRBF_MEGA_TAIL": "20"},
    {"MRBF_MEGA_SHALF": "0"},                                                 # streamed tiles as 128-row jobs everywhere
    {"MRBF_MEGA_SHALF": "12", "MRBF_MEGA_SHALF_HEAD": "3", "MRBF_MEGA_RESERVE": "40", "MRBF_MEGA_XCHAIN": "2"},  # halves at the edges only
    {"MRBF_MEGA_SHALF": "99", "MRBF_MEGA_CHAIN": "96", "MRBF_MEGA_XCHAIN": "8"},
    {"MRBF_MEGA_SHALF": "99", "MRBF_MEGA_CHAIN": "40", "MRBF_MEGA_XCHAIN": "0"},
    {"MRBF_MEGA_TFULL": "0"},                                                 # every panel tile as one 128-row job
    {"MRBF_MEGA_TFULL": "3", "MRBF_MEGA_SHALF": "0"},                         # only those more than three block rows below the streamed ones
]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: ",".join("%s=%s" % (k[10:], v) for k, v in e.items()) or "defaults")
def test_schedule_variant_factors_correctly(env):
    e = dict(os.environ)
    e.update(env)
    e["POTRF_CHECK"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "potrf_time.py"), "2048,6400", "2"], env=e, cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("n=")][-1]
    parts = line.split(" | ")
    assert len(parts) == 2, line
    for part in parts:
        m = re.search(r"info (-?\d+) resid ([0-9.e+-]+)", part)
        assert m, part
        assert int(m.group(1)) == 0, part
        assert float(m.group(2)) < 1e-13, part


def test_fit_front_ends_agree(tmp_path):
    """The fit's two front ends for d <= 64 -- tail basis, Q1'Y and projected right-hand sides in three launches (small.hip, TailQ,
    the default) against the twelve-launch chain it replaces (MRBF_TAILQ=0) -- give the same models: weights and tail coefficients
    agree to rounding-times-conditioning, both interpolate (eight shapes: d = 3 .. 64, k = 1 .. 16, n = 513 .. 3000 incl. non-multiples of 16).  (The switch is read once per process: one child per setting.)"""
    import numpy as np
    outs = []
    for v in ("1", "0"):
        e = dict(os.environ)
        e["MRBF_TAILQ"] = v
        f = str(tmp_path / ("fit%s.npz" % v))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fit_dump.py"), f], env=e, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        outs.append(np.load(f))
    a, b = outs
    for ci in range(8):
        assert int(a["path%d" % ci]) == 2 and int(b["path%d" % ci]) == 2  # the projected Cholesky path in both
        assert float(a["res%d" % ci]) < 1e-7 and float(b["res%d" % ci]) < 1e-7, (ci, float(a["res%d" % ci]), float(b["res%d" % ci]))
        for key in ("w", "lam"):
            x, y = a["%s%d" % (key, ci)], b["%s%d" % (key, ci)]
            assert np.abs(x - y).max() <= 1e-6 * np.abs(y).max(), (ci, key, np.abs(x - y).max(), np.abs(y).max())
