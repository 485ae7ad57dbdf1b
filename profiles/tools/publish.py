"""Copy the judged summaries of a checkpoint (profiles/tools/checkpoint.sh TAG) from gpurun_out/ into profiles/:
kernel-time summaries of the default bench command and of the C4 slice, per-kernel PMC means, and the small JSONs bench.py
reads for `traffic`.  usage: python profiles/tools/publish.py TAG ROUND   (e.g. r02f r02)"""
import glob, json, os, shutil, subprocess, sys
tag, rnd = sys.argv[1:3]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for cfg, name in (("c2", "default_bench"), ("c4", "c4slice_128rx_61M44")):
    f = glob.glob("%s/%s_%s/*/*kernel_stats.csv" % (G, tag, cfg))[0]
    shutil.copy(f, "%s/%s_kernel_stats_%s.csv" % (P, rnd, name))
    txt = subprocess.run([sys.executable, os.path.join(P, "tools", "kstats.py"), "%s/%s_%s" % (G, tag, cfg), "60"], capture_output=True, text=True).stdout
    open("%s/%s_kernel_stats_%s.txt" % (P, rnd, name), "w").write(txt)
shutil.copy("%s/%s_bench.json" % (G, tag), "%s/%s_bench_default.json" % (P, rnd))
def pmc(cfg, *subs):
    return json.loads(subprocess.run([sys.executable, os.path.join(P, "tools", "pmcsum.py"), G, tag, cfg] + list(subs), capture_output=True, text=True).stdout)
src = "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC_HIT/MISS, SQ_*), one pass per counter block, profiles/tools/pmcset.sh"
for cfg in ("c2", "c4", "c5"):
    d = pmc(cfg, "k_frontend_wave", "k_frontend_pk", "k_fir_tiled<3", "k_fir_tiled<5", "k_fir_tiled<2", "k_agc_gain", "k_rxf_tiled")
    json.dump(d, open("%s/%s_pmc_%s.json" % (P, rnd, cfg), "w"), indent=1)
    fe = d.get("k_frontend_pk") if cfg == "c5" and d.get("k_frontend_pk") else d.get("k_frontend_wave")
    if fe:
        out = {"source": src + "; summary in profiles/%s_pmc_%s.json" % (rnd, cfg), "kernel": fe["kernel"]}
        for k in ("hbm_bytes_per_launch", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "l2_hit_rate"):
            if k in fe: out[k] = fe[k]
        json.dump(out, open("%s/%s_pmc_frontend_%s.json" % (P, rnd, cfg), "w"), indent=1)
d = pmc("fft", "k_fft_cols", "k_fft_rows", "k_graph_release")
json.dump(d, open("%s/%s_pmc_fft_kernels.json" % (P, rnd), "w"), indent=1)
if "k_fft_cols" in d and "k_fft_rows" in d:
    cols, rows = d["k_fft_cols"]["hbm_bytes_per_launch"], d["k_fft_rows"]["hbm_bytes_per_launch"]
    rel = d.get("k_graph_release", {}).get("hbm_bytes_per_launch", 0.0)
    json.dump({"source": "rocprofv3 --pmc passes over profiles/tools/fftbench.py (8 frames per launch pair), profiles/tools/pmcset.sh; per-kernel means in profiles/%s_pmc_fft_kernels.json" % rnd,
               "hbm_bytes_per_frame": (cols + rows + rel) / 8.0, "release_bytes_per_frame": rel / 8.0, "algorithmic_bytes_per_frame": 16 << 20,
               "k_fft_cols_bytes_per_launch_of_8_frames": cols, "k_fft_rows_bytes_per_launch_of_8_frames": rows},
              open("%s/%s_pmc_fft.json" % (P, rnd), "w"), indent=1)
print("published", tag, "->", rnd)
