"""Spectrum-only loop for profiling (rocprofv3 -- python profiles/tools/fftbench.py [frames_per_block] [groups])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _hip import DevBuf, sync
from quisk_amd import Engine
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = 1 << 20
rng = np.random.default_rng(3)
x = np.round((rng.standard_normal(16 * N) + 1j * rng.standard_normal(16 * N)) * 2.0 ** 20)
dev = DevBuf(x)
eng = Engine(61440000, n_receivers=0, max_block=8 * N, fft_size=N, data_width=4096)
def run(groups):
    for g in range(groups):
        if F == 8:
            eng.process_device(dev.at((g % 2) * 8 * N * 16), 8 * N)
        else:
            for i in range(8 // F):
                eng.process_device(dev.at(((g % 2) * 8 + i * F) * N * 16), F * N)
        eng.get_graph(1.0, 0.0)
run(2); sync()
t0 = time.perf_counter(); run(G); sync(); dt = time.perf_counter() - t0
print("frames per block %d: %.0f frames/s, %.2f us per frame" % (F, 8 * G / dt, dt / (8 * G) * 1e6))
eng.close()
