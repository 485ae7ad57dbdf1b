"""Does the headline leg speed up once the GPU has been busy?  The bench's timed leg (W warm-up, K timed steps) run five times in one process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from quisk_amd.engine import KT_FRONTEND, KT_AGC
from quisk_amd.synth import make_iq, tune_offsets, mode_mix
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fs, nrx, block = 1536000, 64, 1 << 20
tunes, modes = tune_offsets(nrx, fs), mode_mix(nrx, "ssb")
eng = bench.make_engine(fs, nrx, block, 0, tunes, modes)
nb = 16
base = make_iq(block, fs, tunes[::4][:16], modes[::4][:16], 1002)
host = np.empty((nb, block), dtype=np.complex128)
for b in range(nb): host[b] = np.round(base * np.exp(1j * 0.37 * b))
dev = torch.from_numpy(host.view(np.float64).reshape(nb, block, 2)).cuda()
ptrs = [dev[b].data_ptr() for b in range(nb)]
for rep in range(5):
    dt, st = bench.timed_leg(eng, ptrs, block, K, 5, torch.cuda.synchronize, (KT_FRONTEND, KT_AGC) if rep >= 3 else (KT_FRONTEND,))
    print("run %d: %.4f ms per step, fe %.3f, agc %.3f" % (rep, dt / K * 1e3, st.kernel_ms_sum[KT_FRONTEND] / max(1, st.kernel_launches[KT_FRONTEND]), st.kernel_ms_sum[KT_AGC] / max(1, st.kernel_launches[KT_AGC])))
    if rep == 2: time.sleep(0.5); print("(slept 0.5 s)")
tiles = eng.stats().blocks * 0 + 5 * (K + 5) * (32768 // 64)
pp = [eng.L.qk_debug_agc_passes(eng.h, r) for r in range(nrx)]
print("speculation passes per 64-sample tile: min %.3f mean %.3f max %.3f" % (min(pp) / tiles, sum(pp) / len(pp) / tiles, max(pp) / tiles))
eng.close()
