"""gpurun_out/r02prof_b$B (profiles/collect_r02.sh) -> the committed summaries profiles/r02_{bench,kernel_stats,pmc}_b$B.*
    python experiments/profile_to_json.py B
"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = int(sys.argv[1])
src = os.path.join(ROOT, "gpurun_out", "r02prof_b%d" % B)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "r02_kernel_stats_b%d.csv" % B))
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(dst, "r02_bench_b%d.json" % B), "w"), indent=1)
pmc = json.load(open(os.path.join(src, "pmc.json")))
hot = [k for k in pmc["kernels"] if k.startswith("k_lif_seq_c32d")][0]
out = {"command": "rocprofv3 --pmc <one group per pass> --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                  "--batch %d --cpu-windows 0 (T=128; passes: FETCH_SIZE | WRITE_SIZE | SQ_INSTS_MFMA SQ_INSTS_VALU "
                  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                  "SQ_WAVE_CYCLES SQ_INSTS_LDS); profiles/collect_r02.sh, merged by experiments/pmc_summary.py" % B,
       "units": pmc["units"],
       "hbm_correction": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE counts wide coalesced reads at half "
                         "(MI355X_MICROARCH.md, rocprofv3 HBM section)",
       "k_lif_seq_c32_batch": B if B <= 6144 else None, "k_lif_seq_c32_kernel": hot,
       "k_lif_seq_c32_traffic_bytes_per_launch": pmc["kernels"][hot]["hbm_bytes_per_launch"],
       "note": "batch 8192 runs as chunks of 6144 + 2048 windows under the 24 GB pv budget: per-launch numbers are averages "
               "over both chunk sizes" if B > 6144 else "",
       "kernels": pmc["kernels"]}
json.dump(out, open(os.path.join(dst, "r02_pmc_b%d.json" % B), "w"), indent=1)
print("wrote profiles/r02_*_b%d" % B)
