#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc SQ pass into matrix-pipe / vector-issue utilisation per bench.py kernel class
("rocprof ... MFMA utilisation" of the headline configuration).

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES \\
            --kernel-trace --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 3 --warmup 1 \\
            --no-cpu-baseline --no-graph --profile-steps 0 --resident-inputs
  python profiles/make_mfma.py gpurun_out/pmc_sq profiles/pmc_mfma_by_class.json

Units (MI355X_MICROARCH.md, price list): SQ_VALU_MFMA_BUSY_CYCLES counts CYCLES of matrix-pipe occupancy summed over the
chip's SIMDs (16 per v_mfma_f32_16x16x32_bf16); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_VALU / SQ_WAIT_* count QUAD-cycles summed
over waves.  Reported per class:
  mfma_busy_of_chip   = MFMA busy cycles / (1024 SIMDs x launch duration x clock)     - the matrix pipes' duty cycle
                        (clock: 2.4 GHz nominal; the chip runs lower under load, so this is a lower bound on the duty cycle)
  mfma_busy_of_waves  = MFMA busy cycles / (4 x SQ_WAVE_CYCLES)                        - per resident wave-cycle
  valu_active_of_waves, wait_any_of_waves, wait_inst_of_waves: the same denominators
Launch durations come from the kernel trace of the SAME (profiled, eager) run: longer than under graph replay."""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_traffic import classify  # noqa: E402

CLOCK_HZ = 2.4e9
N_SIMD = 1024


def main(d, out_path):
    cc = glob.glob(d + "/*/*_counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*_kernel_trace.csv")
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(cc)):
        c = classify(r["Kernel_Name"])
        if not c:
            continue
        acc[c][r["Counter_Name"]] += float(r["Counter_Value"])
        did = r.get("Dispatch_Id")
        if did not in seen[c]:
            seen[c].add(did)
            acc[c]["_dur"] += dur.get(did, 0.0)
    out = {}
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        import bench
        out["_build_id"] = bench.build_id()
    except Exception as e:   # noqa: BLE001
        out["_build_id"] = None
        print("warning: no build id:", e, file=sys.stderr)
    for c, v in acc.items():
        n = len(seen[c])
        wave = v.get("SQ_WAVE_CYCLES", 0.0)
        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        e = {"launches_sampled": n, "avg_launch_us_profiled": round(1e6 * v["_dur"] / n, 2) if v["_dur"] else None,
             "mfma_busy_cycles_per_launch": mf / n, "wave_quad_cycles_per_launch": wave / n}
        if v["_dur"]:
            e["mfma_busy_of_chip"] = round(mf / (N_SIMD * v["_dur"] * CLOCK_HZ), 4)
        if wave:
            e["mfma_busy_of_waves"] = round(mf / (4 * wave), 4)
            e["valu_active_of_waves"] = round(v.get("SQ_ACTIVE_INST_VALU", 0.0) / wave, 4)
            e["wait_any_of_waves"] = round(v.get("SQ_WAIT_ANY", 0.0) / wave, 4)
            e["wait_inst_of_waves"] = round(v.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4)
        out[c] = e
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
