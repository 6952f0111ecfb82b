"""bench.py's per_step_paths record (timesteps back to back, HIP-event device time) under the control knobs:
    DCLL_STEP_RO_MULTI=0            per-layer readout tails instead of dcll_step_readouts_multi
    DCLL_BWD_MULTI_MAX_BATCH=0|512  never / also at B = 512 the joint dv launch of dcll_conv_lif_backward_open_multi"""


def main():
    import os, sys
    sys.path.insert(0, os.getcwd())
    import torch, bench
    r = bench.per_step_paths(torch.device("cuda", 0))
    print("bwd_multi_max_batch", os.environ.get("DCLL_BWD_MULTI_MAX_BATCH", "128"), "ro_multi", os.environ.get("DCLL_STEP_RO_MULTI", "1"),
          "test dev", ["%.4f" % v for v in r["test_device_ms_per_timestep_all"]],
          "learn dev", ["%.4f" % v for v in r["learn_device_ms_per_timestep_all"]],
          "learn wall", ["%.4f" % v for v in r["learn_wall_ms_per_timestep_all"]])


if __name__ == "__main__":
    main()
