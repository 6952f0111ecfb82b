"""gpurun_out/<round>prof_* (profiles/collect_%s.sh) -> the committed summaries profiles/<round>_{bench,kernel_stats,pmc}_*.*
    python experiments/profile_to_json.py r03 4096 | ref | plane128
"""


def main():
    import json, os, shutil, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rnd, mode = sys.argv[1], sys.argv[2]
    tag = {"ref": "ref_b4096", "plane128": "plane128_b64", "t1024": "t1024_b512"}.get(mode, "b%s" % mode)
    src = os.path.join(ROOT, "gpurun_out", "%sprof_%s" % (rnd, mode if mode in ("ref", "plane128", "t1024") else "b" + mode))
    dst = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_%s.csv" % (rnd, tag)))
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
    json.dump(bench, open(os.path.join(dst, "%s_bench_%s.json" % (rnd, tag)), "w"), indent=1)
    pmc = json.load(open(os.path.join(src, "pmc.json")))
    hot_prefix = {"ref": "k_lif_seq_w3<64", "plane128": "k_lif_seq_c32t"}.get(mode, "k_lif_seq_c32d")
    hot = [k for k in pmc["kernels"] if k.startswith(hot_prefix)]
    out = {"command": "rocprofv3 --pmc <one group per pass> --output-format csv -- python3 bench.py <args of profiles/collect_%s.sh %s> "
                      "--steps 1 --warmup 0 (T=128; passes: FETCH_SIZE | WRITE_SIZE | SQ_INSTS_MFMA SQ_INSTS_VALU "
                      "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                      "SQ_WAVE_CYCLES SQ_INSTS_LDS); merged by experiments/pmc_summary.py" % (rnd, mode),
           "units": pmc["units"],
           "hbm_correction": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB: gfx950 FETCH_SIZE counts wide coalesced reads at half "
                             "(MI355X_MICROARCH.md, rocprofv3 HBM section)",
           "hot_kernels": hot, "kernels": pmc["kernels"]}
    if hot_prefix == "k_lif_seq_c32d" and hot:
        out.update(k_lif_seq_c32_batch=512 if mode == "t1024" else int(mode), k_lif_seq_c32_kernel=hot[0],
                   k_lif_seq_c32_traffic_bytes_per_launch=pmc["kernels"][hot[0]]["hbm_bytes_per_launch"])
    json.dump(out, open(os.path.join(dst, "%s_pmc_%s.json" % (rnd, tag)), "w"), indent=1)
    print("wrote profiles/%s_*_%s" % (rnd, tag))


if __name__ == "__main__":
    main()
