"""How often does a repetition of bench.per_step_paths come out slow, on which clock, and is it the garbage collector?
(round-3 driver run: learn 1.99 ms per timestep against 0.73-0.75 in every builder-side run; round 4: 1 of 72 repetitions of
`test` at 1.88 ms wall = a ~70 ms stall.)  Records every collection of Python's cyclic GC with its duration.
usage: python experiments/per_step_outliers.py [reps] [gc: on|off|freeze]"""


def main():
    import gc
    import json
    import os
    import sys
    import time

    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench  # noqa: E402

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    mode = sys.argv[2] if len(sys.argv) > 2 else "on"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    events, t_start = [], [0.0]


    def cb(phase, info):
        if phase == "start":
            t_start[0] = time.perf_counter()
        else:
            events.append((info["generation"], 1e3 * (time.perf_counter() - t_start[0])))


    gc.callbacks.append(cb)
    if mode == "off":
        gc.disable()
    elif mode == "freeze":
        gc.collect()
        gc.freeze()
    for trial in range(3):
        del events[:]
        r = bench.per_step_paths(dev, reps=reps)
        for k in ("test_wall_ms_per_timestep_all", "test_device_ms_per_timestep_all", "learn_wall_ms_per_timestep_all",
                  "learn_device_ms_per_timestep_all"):
            print(trial, k, " ".join("%.3f" % v for v in r[k]), flush=True)
        by_gen = {g: [d for g_, d in events if g_ == g] for g in (0, 1, 2)}
        print(trial, "gc", mode, {g: (len(v), round(max(v), 2) if v else 0) for g, v in by_gen.items()},
              "objects", len(gc.get_objects()), flush=True)


if __name__ == "__main__":
    main()
