"""Diagnostic: per-phase wave-cycle shares of k_cfconv_fused (needs `make -C agdiff_amd/csrc clean all EXTRA=-DAG_CONV_STAMPS`)."""
import ctypes, json, subprocess, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["bench.py", "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-traj"] + sys.argv[1:]
import runpy
from agdiff_amd import _lib
lib = ctypes.CDLL(_lib.LIBPATH)
buf = (ctypes.c_ulonglong * 8)()
try:
    runpy.run_path("bench.py", run_name="__main__")
finally:
    lib.agdiff_debug_conv_stamps(buf, 1)
    v = list(buf)
    tot = sum(v[:4])
    names = ["meta+flush", "layer1+softplus+split", "bounds+masks", "layer2+message+reduce"]
    print("waves", v[7])
    for n, x in zip(names, v[:4]):
        print("%-16s %6.2f %%   %.0f cycles/wave-launch" % (n, 100.0 * x / max(tot, 1), x / max(v[7], 1)))
