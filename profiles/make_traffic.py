#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, as MI355X_MICROARCH.md
prescribes) into HBM bytes per launch per bench.py kernel class.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py ... (same)
  python profiles/make_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic_by_class.json \
         [kernel_stats.csv of a `rocprofv3 --kernel-trace --stats` run under hipGraph replay]

The output is stamped with `_build_id` (bench.build_id(): sha-256 over csrc/) so that bench.py quotes it only for the
build it was taken from, and - when the stats CSV is given - carries each class's average launch duration under graph
replay (`avg_launch_us_graph`) and the CSV's name (`_rocprof_stats`).

gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE is in KB and reports exactly half of the bytes of 16-byte-per-lane
reads -> x2; WRITE_SIZE (KB) is exact for 16-byte-per-lane stores and float atomics.  bytes = (2*FETCH + WRITE) * 1024
(`hbm_bytes_per_launch`).  Round 5 calibrated the two access shapes of the weight-gradient workgroups that rounds 3-4 had
left open (scratch/fetch_calib.hip, profiles/r05_fetch_calibration.txt): 128-byte row segments of panels whose rows are
512 / 1024 / 2048 bytes apart read FETCH_SIZE x 1024 = 0.500 of the known bytes, exactly like a contiguous stream; 64-byte
fp32 atomic-add segments read WRITE_SIZE = the bytes added and FETCH_SIZE = 0.  So x2 holds for every class of the step
and `hbm_bytes_per_launch` is ONE number (the x1 figure of round 4's files is gone).
"""
import collections, csv, glob, json, re, sys

EPI = {0: "store", 1: "store_f32", 2: "resid", 3: "gelu", 4: "scale_resid", 5: "glu", 6: "glu_bwd", 7: "gelu_bwd", 8: "store_res"}


def classify(name):
    m = re.search(r"gemm_nt_kernelIDF16bLi(\d+)ELi(\d+)ELi\d+E", name)
    if m:
        return f"gemm_nt<bf16,{EPI[int(m.group(1))]},{m.group(2)}>"
    m = re.search(r"gemm_nt_kernel<.*ELi(\d+)E", name)   # demangler-confused names: GLU_BWD (Li6)
    if m:
        return f"gemm_nt<bf16,glu_bwd,{m.group(1)}>"
    if "ffn_fwd_fused_kernel" in name and "ffn_qkv" not in name:
        return "attn_out_fused" if "Lb0" in name else "ffn_fwd_fused"
    for key, cls in (("ffn_fwd_big_kernel", "ffn_fwd_big"), ("adamw_norm_kernel", "adamw_norm"), ("adamw_commit_kernel", "adamw_commit"), ("adamw_persist_kernel", "adamw_persist"),
                     ("ffn_qkv_fwd_fused_kernel", "ffn_qkv_fwd_fused"), ("tlayer_fwd_kernel", "tlayer_fwd"), ("ffn_bwd_fused_kernel", "ffn_bwd_fused+wgrad"),
                     ("block_run_fwd_kernel", "block_run_fwd"),
                     ("gemm_tn_group_kernel", "gemm_tn_wgrad_group<bf16>"), ("dwblock_bwd_kernel", "dwblock_bwd"),
                     ("block_fwd_fused_kernel", "block_fwd_fused"), ("block_bwd_fused_kernel", "block_bwd_fused+wgrad"),
                     ("qkv_fwd_fused_kernel", "qkv_fwd_fused"), ("qkv_bwd_fused_kernel", "qkv_bwd_fused+wgrad"),
                     ("lnout_bwd_fused_kernel", "lnout_bwd_fused+wgrad"), ("block_small_fwd_kernel", "block_small_fwd"),
                     ("block_small_bwd_kernel", "block_small_bwd"), ("block_small_bwd_full_kernel", "block_small_bwd_full"),
                     ("attn_global_fwd_fp8_kernel", "attn_global_fwd_fp8"),
                     ("attn_global_bwd_kernel", "attn_global_bwd"), ("attn_global_fwd_kernel", "attn_global_fwd"),
                     ("attn_local_bwd_kernel", "attn_local_bwd"), ("attn_local_fwd_kernel", "attn_local_fwd"),
                     ("ln_bwd_kernel", "layernorm_bwd"), ("ln_fwd_kernel", "layernorm_fwd"),
                     ("cnn_gamma_finish_kernel", "cnn_gamma_finish"), ("reduce_partials_kernel", "reduce_partials"),
                     ("convert_batch_kernel", "convert_batch"), ("gemm_tn_kernel", "gemm_tn_wgrad<bf16>"),
                     ("dwln_fwd_kernel", "dwln_fwd"), ("stem_fwd_kernel", "stem_fwd"), ("stem_bwd_kernel", "stem_bwd")):
        if key in name:
            return cls
    return None


def load(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        c = classify(r["Kernel_Name"])
        if c:
            acc[c][0] += float(r["Counter_Value"]); acc[c][1] += 1
    return acc


def total(d, counter):
    """sum of the counter over EVERY kernel of the run (classified or not), and the number of profiled launches"""
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    t, n, steps = 0.0, 0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            t += float(r["Counter_Value"]); n += 1
            steps += "bce_kernel" in r["Kernel_Name"]          # one launch per training step
    return t, n, steps


def graph_durations(stats_csv):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(stats_csv)):
        c = classify(r["Name"])
        if c:
            acc[c][0] += float(r["TotalDurationNs"]); acc[c][1] += int(r["Calls"])
    return {c: v[0] / v[1] / 1e3 for c, v in acc.items() if v[1]}


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    try:
        import bench
        out["_build_id"] = bench.build_id()
    except Exception as e:   # noqa: BLE001
        out["_build_id"] = None
        print("warning: no build id:", e, file=sys.stderr)
    gd = {}
    if len(sys.argv) > 4:
        gd = graph_durations(sys.argv[4])
        out["_rocprof_stats"] = "profiles/" + os.path.basename(sys.argv[4])
    for c in fe:
        if c in wr and fe[c][1] == wr[c][1]:
            n = fe[c][1]
            out[c] = {"launches_sampled": n, "fetch_size_kb_raw_per_launch": fe[c][0] / n,
                      "write_size_kb_per_launch": wr[c][0] / n,
                      "hbm_bytes_per_launch": (2 * fe[c][0] + wr[c][0]) * 1024 / n,
                      "fetch_correction": "x2 (16-byte-per-lane reads incl. 128-byte row segments; atomics fetch nothing: profiles/r05_fetch_calibration.txt)"}
            if c in gd:
                out[c]["avg_launch_us_graph"] = round(gd[c], 2)
    # the whole run (bench.py --steps 3 --warmup 1 --no-graph: 4 training steps + the eager preparation runs; divide by the
    # steps the caller knows it ran): every kernel, classified or not
    ft, fn, fs = total(sys.argv[1], "FETCH_SIZE")
    wt, wn, ws = total(sys.argv[2], "WRITE_SIZE")
    out["_run_total"] = {"fetch_size_kb_raw": ft, "write_size_kb": wt, "launches": fn, "training_steps": fs,
                         "hbm_bytes": (2 * ft + wt) * 1024, "hbm_bytes_per_step": (2 * ft + wt) * 1024 / max(fs, 1),
                         "hbm_bytes_written_per_step": wt * 1024 / max(fs, 1)}
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))
