#!/usr/bin/env python3
"""Writes the regression fixtures of this directory from the CPU oracle (oracle/slb_oracle.c).

These are NOT outputs of the reference program (it cannot be built here: it needs GSL);
the one recorded reference output is config1_display4_reference.json.  They pin the
oracle -- and through the GPU tests the kernels -- against drift:

  fields_N8_M61.npz      all nine arrays after the tiptoe and after 1, 2 and 10 loop iterations
  av_N8_M61.npz          av_data after 1, 2, 3 calls of av() on the state after 10 iterations
  display77_shrunken_config2.txt   the display=77 trace of n-harmonics=64 g-grid=32768 (11 records)
  step_counts.json       loop iterations for a few (dt, t-max, omega): float accumulation of t

    python tests/golden/make_fixtures.py
"""
import ctypes as C
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402

FIELDS = dict(N=8, M=61, B=0.05, omega=100.0)


def fields():
    p = ob.make_params(**FIELDS)
    st = ob.OracleState(p, threads=1)
    out, done = {}, 0
    for steps in (0, 1, 2, 10):
        st.step(steps - done)
        done = steps
        out[f"eq_{steps}"] = st.array("a0").copy()          # a0 (the key a0_<k> is a[0])
        for i in range(4):
            out[f"a{i}_{steps}"] = st.array("a", i).copy()
            out[f"b{i}_{steps}"] = st.array("b", i).copy()
        out[f"idx_{steps}"] = np.array([st.st.current, st.st.next, st.st.current_hs, st.st.next_hs])
        out[f"t_{steps}"] = np.float32(st.st.t)
    np.savez_compressed(os.path.join(HERE, "fields_N8_M61.npz"), **out)
    cur = st.st.current
    a, b = st.array("a", cur), st.array("b", cur)
    av = np.zeros(6, np.float32)
    avs = {}
    for k, t in enumerate((0.2, 0.201, 0.202), 1):
        ob.lib().orc_av(C.byref(p), a, b, av, np.float32(t))
        avs[f"after_{k}"] = av.copy()
    np.savez(os.path.join(HERE, "av_N8_M61.npz"), times=np.array([0.2, 0.201, 0.202], np.float32), **avs)
    st.close()


def display77():
    args = ["display=77", "omega=100", "B=0.12", "n-harmonics=64", "g-grid=32768", "t-max=0.05", "E_dc=1",
            "E_omega=0.1", "mu=5", "alpha=1", "PhiYmin=-4", "PhiYmax=4", "dt=0.001", "o=out.txt"]
    with tempfile.TemporaryDirectory() as d:
        ob.run_cli(args, cwd=d)
        text = open(os.path.join(d, "out.txt")).read()
    open(os.path.join(HERE, "display77_shrunken_config2.txt"), "w").write(text)


def step_counts():
    cases = []
    for dt, tmax, omega in ((0.001, 10.0, 1.0), (0.001, 0.05, 100.0), (0.001, 0.001, 100.0), (0.0005, 0.1, 50.0)):
        f = np.float32
        T = f(2 * 3.141592653589793115998 / np.float64(f(omega)))
        t_max, t, n = f(f(tmax) + T), f(0), 0
        while t < t_max:
            t = f(t + f(dt))
            n += 1
        cases.append({"dt": dt, "t-max": tmax, "omega": omega, "iterations": n, "t_max": "%0.20f" % t_max})
    json.dump(cases, open(os.path.join(HERE, "step_counts.json"), "w"), indent=1)


if __name__ == "__main__":
    fields()
    display77()
    step_counts()
    print("fixtures written to", HERE)
