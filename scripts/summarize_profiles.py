#!/usr/bin/env python
"""Turn gpurun_out/prof/* (scripts/collect_profiles.sh) into the committed summaries under profiles/:

  profiles/rNN_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py`
  profiles/rNN_pmc_summary.md           HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes,
                                        with the calibration against the plain-copy kernel
  profiles/pmc_traffic.json             what bench.py reports as roofline.traffic

Counter handling follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected
in separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE reads exactly 1/2 of a wide coalesced
read stream -- the factor is re-measured here on a copy of known size rather than assumed.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.environ.get("TPSPP_PROFILE_DST") or os.path.join(ROOT, "profiles")   # (the GPU box writes to gpurun_out/prof_summary)
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
KERNEL = os.environ.get("TPSPP_PROFILE_KERNEL", "tps_warp_pair_kernel")
COPY_BYTES = 512 * 3 * 32 * 100 * 4          # scripts/ubench/copy_bench.hip: bytes read = bytes written
ALGO_BYTES = 512 * 76960


def mean_counter(path, kernel_substr, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel_substr in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(vals) / len(vals), len(vals)


def mfma_summary():
    """MFMA-busy fraction per kernel of the TPS++ module run: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128),
    the denominator calibrated on a kernel that does nothing but fp32 MFMAs (scripts/ubench/mfma_bench.hip)."""
    def load(path):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        return agg
    mod = os.path.join(SRC, "mfma_pmc", "module_counter_collection.csv")
    cal = os.path.join(SRC, "mfma_cal", "cal_counter_collection.csv")
    if not (os.path.exists(mod) and os.path.exists(cal)):
        return
    c = next(v for k, v in load(cal).items() if "mfma_only" in k)
    busy, act = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
    per_active = busy / act                       # counter units of a saturated matrix pipe per GUI_ACTIVE tick
    tf = [l for l in open(os.path.join(SRC, "mfma_cal.log")) if "TFLOP/s" in l]
    rows = []
    for k, v in load(mod).items():
        b = v.get("SQ_VALU_MFMA_BUSY_CYCLES")
        a = v.get("GRBM_GUI_ACTIVE")
        if not b or not a or sum(b) == 0:
            continue
        rows.append((sum(b), k, len(b), sum(b) / sum(a) / per_active))
    rows.sort(reverse=True)
    with open(os.path.join(DST, f"{TAG}_mfma_util.md"), "w") as f:
        f.write(f"""# {TAG}: matrix-pipe utilisation of the TPS++ regressor kernels (MI355X, 1 GPU)

`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 scripts/bench_module.py` (batch 512, own pass; the
script runs the exact-fp32, the "bf16x3" (`..., true>` instantiations) and the bf16 configuration one after the other,
so the first table holds the kernels of all three; per-configuration passes follow).

Calibration (`scripts/ubench/mfma_bench.hip`: every SIMD issues only `v_mfma_f32_32x32x2_f32`):
{tf[-1].strip() if tf else ""}
-> SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE = {per_active:.2f} at saturation (64 busy cycles per MFMA, 1024 SIMDs,
GRBM_GUI_ACTIVE summed over the 8 XCDs); utilisation below = the same ratio of a kernel / {per_active:.2f}.

| kernel | dispatches | MFMA pipe busy |
|---|---|---|
""")
        for _, k, n, u in rows:
            f.write(f"| `{k[:100]}` | {n} | {100 * u:.1f} % |\n")
        f.write("\n(counter collection serialises dispatches and slows kernels slightly; un-profiled rates are in DESIGN.md section 4b)\n")
        # the bf16 and "bf16x3" configurations of the same module: the pipe-busy ratio does not depend on the operand type
        for mode, title in (("bf16only", "bf16 configuration (`bench_module.py 512 bf16only`)"),
                            ("x3only", '"bf16x3" configuration (`bench_module.py 512 x3only`): fp32 tensors, three bf16 MFMAs per product'),
                            ("backbone", "bf16 backbone + TPS++ (`scripts/debug/bench_backbone.py bf16`, batch 512: `conv3_wide_kernel` = the "
                                         "128- to 512-channel 3x3 layers, `conv_stem_bf16_kernel` = the stem; round 6)"),
                            ("wide", "the backbone's wide 3x3 layers alone, 35 launches back to back each (`scripts/debug/bench_wide.py both`): "
                                     "`conv3_wide_kernel` against the tiled kernel it replaces (`conv_tiled_bf16_kernel<3, 1, 1, 4, 32, ...>` = "
                                     "8x32 maps, `<3, 1, 1, 4, 16, 2, ...>` = 4x16 maps)")):
            pth = os.path.join(SRC, f"mfma_pmc_{mode}", "module_counter_collection.csv")
            if not os.path.exists(pth):
                continue
            rows2 = []
            for k, v in load(pth).items():
                b, a = v.get("SQ_VALU_MFMA_BUSY_CYCLES"), v.get("GRBM_GUI_ACTIVE")
                if not b or not a or sum(b) == 0:
                    continue
                rows2.append((sum(b), k, len(b), sum(b) / sum(a) / per_active))
            rows2.sort(reverse=True)
            f.write(f"\n## {title}\n\n| kernel | dispatches | MFMA pipe busy |\n|---|---|---|\n")
            for _, k, n, u in rows2:
                f.write(f"| `{k[:100]}` | {n} | {100 * u:.1f} % |\n")


def timeline(trace_csv, kernel_substr, steps):
    """From a rocprofv3 kernel trace: the last `steps` dispatches of the warp kernel (the timed region; the one-stream
    re-run of bench.py follows it on stream 0, so the region is found by its streams) -> average start-to-end duration
    of a kernel, and the period between launches = (last end - first start) / launches."""
    rows = [r for r in csv.DictReader(open(trace_csv)) if kernel_substr in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def line_regions(bench):
    """(S, regions_us on S streams, regions_us on one stream) of a bench line; round 5 made them flat strings of
    `roofline` (the driver's parser drops nested objects), earlier lines carry nested lists."""
    rf = bench["roofline"]
    if "multi_stream_streams" in rf:
        S = int(rf["multi_stream_streams"])
        f = lambda k: [float(x) for x in str(rf.get(k, "")).split()] or None       # noqa: E731
        return S, f("regions_us_multi_stream"), f("regions_us_one_stream")
    tried = bench["config"].get("streams_tried", [bench["config"]["streams"]])
    return max(tried), rf.get("multi_stream", {}).get("regions_us"), rf.get("one_stream", {}).get("regions_us")


def region_sequence(bench, S, R):
    """(streams, region index) of every timed region in DISPATCH order: round 6 interleaves the protocols (config.region_order),
    earlier lines ran all S-stream regions first."""
    if S <= 1:
        return [(S, r) for r in range(R)]
    if bench.get("config", {}).get("region_order") == "interleaved":
        return [(st, r) for r in range(R) for st in (S, 1)]
    return [(S, r) for r in range(R)] + [(1, r) for r in range(R)]


def protocol_rows(sub, logname):
    """Per-protocol aggregates (streams, dispatches, total, average, min, max, sorted periods) of the warp kernel in one traced
    bench run -- the same split as protocol_tables(), without the per-dispatch file."""
    pth = os.path.join(SRC, sub, "bench_kernel_trace.csv")
    log = os.path.join(SRC, logname)
    if not (os.path.exists(pth) and os.path.exists(log)):
        return []
    lines = [l for l in open(log) if l.startswith("{")]
    if not lines:
        return []
    bench = json.loads(lines[-1])
    steps, warm, R = bench["steps"], bench["warmup"], int(bench["config"].get("repeats", 1))
    S, _, _ = line_regions(bench)
    rows = timeline(pth, KERNEL, steps)
    lo, per = warm, collections.OrderedDict()
    for streams, reg in region_sequence(bench, S, R):
        sel = rows[lo:lo + steps]
        if len(sel) < steps:
            break
        lo += steps
        durs, periods = per.setdefault(streams, ([], []))
        durs += [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in sel]
        periods.append((max(int(x["End_Timestamp"]) for x in sel) - min(int(x["Start_Timestamp"]) for x in sel)) / len(sel))
    return [(streams, len(d), sum(d), sum(d) / len(d), min(d), max(d), sorted(p)) for streams, (d, p) in per.items() if d]


def protocol_tables():
    """The driver's command in the kernel trace, reduced so that the timeline can be recomputed from profiles/ alone:
    rNN_bench_dispatches.csv  -- every dispatch of the warp kernel (protocol, region, queue, start, end; ns from the first)
    and per-protocol aggregate rows for rNN_bench_kernel_stats.csv (one-stream and S-stream dispatches SEPARATED: their
    durations differ by 2x because overlapped kernels share the memory system)."""
    pth = os.path.join(SRC, "trace_driver", "bench_kernel_trace.csv")
    log = os.path.join(SRC, "bench_trace_driver.log")
    if not (os.path.exists(pth) and os.path.exists(log)):
        return []
    lines = [l for l in open(log) if l.startswith("{")]
    if not lines:
        return []
    bench = json.loads(lines[-1])
    steps, warm, R = bench["steps"], bench["warmup"], int(bench["config"].get("repeats", 1))
    S, _, _ = line_regions(bench)
    rows = timeline(pth, KERNEL, steps)
    t0 = int(rows[0]["Start_Timestamp"])
    lo, out, agg = warm, [], []
    for i, r in enumerate(rows[:warm]):
        out.append(("warmup", S, -1, i, r))
    per = collections.OrderedDict()
    for streams, reg in region_sequence(bench, S, R):
        sel = rows[lo:lo + steps]
        if len(sel) < steps:
            break
        for i, r in enumerate(sel):
            out.append(("timed", streams, reg, lo + i, r))
        lo += steps
        durs, periods = per.setdefault(streams, ([], []))
        durs += [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in sel]
        periods.append((max(int(x["End_Timestamp"]) for x in sel) - min(int(x["Start_Timestamp"]) for x in sel)) / len(sel))
    agg = [(streams, len(d), sum(d), sum(d) / len(d), min(d), max(d), sorted(p)) for streams, (d, p) in per.items() if d]
    for i, r in enumerate(rows[lo:]):
        out.append(("after", 1, -1, lo + i, r))
    with open(os.path.join(DST, f"{TAG}_bench_dispatches.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["phase", "streams_of_protocol", "region", "dispatch", "queue_id", "start_ns", "end_ns", "duration_ns"])
        for ph, st, reg, i, r in out:
            a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            wr.writerow([ph, st, reg, i, r.get("Queue_Id", "?"), a, b, b - a])
    return agg


def timeline_summary():
    """Per traced run of bench.py: the dispatch order is warm-up, `repeats` regions of `steps` launches on S streams,
    `repeats` regions on one stream (when S > 1), then the parity step / extras.  For every region: the kernels' average
    start-to-end duration and the period (last end - first start) / launches, next to the bench line's own
    regions_us of the same run."""
    runs = (("the driver's command: `python3 bench.py --gpus 1 --steps 20 --warmup 5`", "trace_driver", "bench_trace_driver.log"),
            ("`--steps 400 --warmup 50` (long regions)", "trace", "bench_trace.log"),
            ("`--steps 400 --warmup 50 --streams 1`", "trace1", "bench_trace1.log"))
    blocks = []
    for tag, sub, logname in runs:
        pth = os.path.join(SRC, sub, "bench_kernel_trace.csv")
        log = os.path.join(SRC, logname)
        if not (os.path.exists(pth) and os.path.exists(log)):
            continue
        lines = [l for l in open(log) if l.startswith("{")]
        if not lines:
            continue
        bench = json.loads(lines[-1])
        steps, warm = bench["steps"], bench["warmup"]
        R = int(bench["config"].get("repeats", 1))
        S, reg_m, reg_1 = line_regions(bench)
        rows = timeline(pth, KERNEL, steps)
        regs_of = {S: reg_m, 1: reg_1 if S > 1 else reg_m}
        lo = warm
        out = []
        for streams, r in region_sequence(bench, S, R):
            regs_us = regs_of[streams]
            sel = rows[lo:lo + steps]
            lo += steps
            if len(sel) < steps:
                break
            dur = [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in sel]
            span = max(int(x["End_Timestamp"]) for x in sel) - min(int(x["Start_Timestamp"]) for x in sel)
            queues = sorted({x.get("Queue_Id", "?") for x in sel})
            overlap = sum(1 for p, q in zip(sel, sel[1:]) if int(q["Start_Timestamp"]) < int(p["End_Timestamp"]))
            out.append((streams, r, len(sel), sum(dur) / len(dur) / 1e3, span / len(sel) / 1e3, len(queues), overlap,
                        regs_us[r] if regs_us and r < len(regs_us) else None))
        blocks.append((tag, bench, out))
    if not blocks:
        return
    with open(os.path.join(DST, f"{TAG}_bench_kernel_timeline.md"), "w") as f:
        f.write(f"""# {TAG}: the timed regions of `bench.py` in the rocprofv3 kernel trace (MI355X, 1 GPU)

bench.py times two launch protocols, each over `repeats` regions of exactly `steps` launches: step i on HIP stream i % S
(S = `config.streams_tried[-1]`; launches on different streams overlap) and all steps on one stream.  Two different numbers
describe a region: a kernel's own start-to-end DURATION (what `--stats` averages) and the PERIOD between launches,
(last end - first start) / launches, which is what `ms_per_step` / `roofline.*.regions_us` of the bench line measure.  On
one stream the two coincide up to the launch gap.  The bench's HIP events bracket a region from the host's first enqueue:
the few microseconds between the start event and the first kernel's start are in the line's figure and not in the trace's
period -- at K = 20 launches per region that is ~0.5 us per launch (5-6 %), at K = 400 the two agree within 1 %.
Computed by scripts/summarize_profiles.py from the
`*_kernel_trace.csv` of each run (the dispatches of `{KERNEL}` in dispatch order: warm-up, then the timed regions -- round 6:
S-stream region r and one-stream region r alternate (`config.region_order`), earlier rounds ran the S-stream regions first).  Profiled runs are slower than un-profiled ones (rocprofv3 adds per-dispatch work), so the trace's
period is compared with the bench line of THE SAME run.

""")
        for tag, bench, out in blocks:
            rf = bench["roofline"]
            one_us = rf.get("one_stream_launch_us", rf.get("one_stream", {}).get("launch_us", float("nan")))
            f.write(f"## {tag}\n\nbench line of this (profiled) run: streams = {bench['config']['streams']}, launch_us = "
                    f"{rf['launch_us']:.2f} (median region), frac = {rf['frac']:.3f}; "
                    f"one stream (median) {one_us:.2f} us = {rf.get('one_stream_frac', float('nan')):.3f}; "
                    f"{rf.get('multi_stream_streams', '?')} streams (median) {rf.get('multi_stream_launch_us_median', float('nan')):.2f} us.\n\n"
                    "| streams | region | launches | average duration, us | period, us | HIP queues | launches that start before the previous one ends | bench line, same region, us |\n"
                    "|---|---|---|---|---|---|---|---|\n")
            for streams, r, n, d, pd, q, ov, lr in out:
                f.write(f"| {streams} | {r} | {n} | {d:.2f} | {pd:.2f} | {q} | {ov} | {'' if lr is None else f'{lr:.2f}'} |\n")
            for streams in sorted({o[0] for o in out}, reverse=True):
                pds = sorted(o[4] for o in out if o[0] == streams)
                drs = [o[3] for o in out if o[0] == streams]
                f.write(f"\n{streams} stream(s): period min {pds[0]:.2f} / median {pds[len(pds) // 2]:.2f} us, "
                        f"average kernel duration {sum(drs) / len(drs):.2f} us.\n")
            f.write("\n")


def main():
    os.makedirs(DST, exist_ok=True)
    timeline_summary()
    # the warp kernel per launch protocol (from the driver-command trace), then rocprofv3's own --stats rows of the long
    # 3-stream run with kernel names cut to 200 characters (PyTorch's RNG kernels have 5-KB names)
    agg = protocol_tables()
    with open(os.path.join(SRC, "trace", "bench_kernel_stats.csv")) as fin, \
            open(os.path.join(DST, f"{TAG}_bench_kernel_stats.csv"), "w", newline="") as fout:
        rd = csv.reader(fin)
        wr = csv.writer(fout, quoting=csv.QUOTE_NONNUMERIC)
        wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "PeriodNs_per_region", "Source"])
        ALG = 39403520                                       # algorithmic bytes per launch
        for streams, n, tot, avg, mn, mx, periods in agg:
            wr.writerow([f"{KERNEL} | timed regions on {streams} stream(s) | {ALG} B / average / 8 TB/s = {ALG / avg / 8000.0:.3f}",
                         n, tot, round(avg, 1), "", mn, mx, " ".join(f"{x:.0f}" for x in periods),
                         "python3 bench.py --gpus 1 --steps 20 --warmup 5 (trace_driver; dispatches in " + f"{TAG}_bench_dispatches.csv); "
                         "20-launch regions: the profiler's per-dispatch work stretches duration and period"])
        # the long regions (400 launches): the kernel's own duration with every launch behind the previous one is the
        # one-stream row of these -- the figure bench.py's roofline.one_stream_frac (un-profiled) is to be compared with
        for sub, logname, cmd in (("trace", "bench_trace.log", "python3 bench.py --steps 400 --warmup 50"),
                                  ("trace1", "bench_trace1.log", "python3 bench.py --steps 400 --warmup 50 --streams 1")):
            for streams, n, tot, avg, mn, mx, periods in protocol_rows(sub, logname):
                wr.writerow([f"{KERNEL} | timed regions on {streams} stream(s) | {ALG} B / average / 8 TB/s = {ALG / avg / 8000.0:.3f}",
                             n, tot, round(avg, 1), "", mn, mx, " ".join(f"{x:.0f}" for x in periods), cmd + " (400-launch regions)"])
        for i, row in enumerate(rd):
            if i == 0:
                continue
            if len(row[0]) > 200:
                row[0] = row[0][:200] + "..."
            wr.writerow([row[0]] + [float(x) if "." in x else int(x) for x in row[1:7]] + ["", "python3 bench.py --steps 400 --warmup 50 (3 streams + 1 stream mixed: see the rows above for the split)"])
    stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(SRC, "trace", "bench_kernel_stats.csv")))}
    krow = next(v for k, v in stats.items() if KERNEL in k)
    bench_line = [l for l in open(os.path.join(SRC, "bench_trace.log")) if l.startswith("{")][-1]
    bench = json.loads(bench_line)

    f_copy, _ = mean_counter(os.path.join(SRC, "cal_FETCH_SIZE", "copy_counter_collection.csv"), "copy_k(", "FETCH_SIZE")
    w_copy, _ = mean_counter(os.path.join(SRC, "cal_WRITE_SIZE", "copy_counter_collection.csv"), "copy_k(", "WRITE_SIZE")
    f_corr = COPY_BYTES / (f_copy * 1024.0)
    w_corr = COPY_BYTES / (w_copy * 1024.0)
    f_k, nf = mean_counter(os.path.join(SRC, "pmc_FETCH_SIZE", "bench_counter_collection.csv"), KERNEL, "FETCH_SIZE")
    w_k, nw = mean_counter(os.path.join(SRC, "pmc_WRITE_SIZE", "bench_counter_collection.csv"), KERNEL, "WRITE_SIZE")
    rd = f_k * 1024.0 * f_corr
    wr = w_k * 1024.0 * w_corr
    traffic = rd + wr
    js = {"round": TAG, "kernel": krow["Name"], "hbm_bytes_per_launch": traffic,
          "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
          "algorithmic_bytes_per_launch": ALGO_BYTES, "traffic_over_algorithmic": traffic / ALGO_BYTES,
          "FETCH_SIZE_KiB_raw": f_k, "WRITE_SIZE_KiB_raw": w_k,
          "fetch_correction_measured_on_copy": f_corr, "write_correction_measured_on_copy": w_corr,
          "dispatches_averaged": [nf, nw]}
    json.dump(js, open(os.path.join(DST, "pmc_traffic.json"), "w"), indent=1)
    with open(os.path.join(DST, f"{TAG}_pmc_summary.md"), "w") as f:
        f.write(f"""# {TAG}: rocprofv3 summary for `python3 bench.py` (MI355X, 1 GPU)

Command lines: `scripts/collect_profiles.sh` (run through gpurun).  Raw CSVs: `gpurun_out/prof/` (scratch).

## Kernel trace (`rocprofv3 --kernel-trace --stats`, file `{TAG}_bench_kernel_stats.csv`)

| kernel | calls | average ns | min ns | max ns |
|---|---|---|---|---|
| `{krow['Name'][:90]}` | {krow['Calls']} | {float(krow['AverageNs']):.0f} | {krow['MinNs']} | {krow['MaxNs']} |

bench.py's own HIP-event figure in the same (profiled) run: launch_us = {bench['roofline']['launch_us']:.2f}
(rocprofv3 serialises dispatches, so profiled runs are slower than the un-profiled bench line).

## HBM traffic (`rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate passes)

Calibration on `copy_k` (scripts/ubench/copy_bench.hip, {COPY_BYTES} B read and {COPY_BYTES} B written per launch):
FETCH_SIZE reads {f_copy:.1f} KiB -> correction x{f_corr:.4f} (the gfx950 "1/2 of a wide coalesced stream" factor),
WRITE_SIZE reads {w_copy:.1f} KiB -> correction x{w_corr:.4f}.

| per launch of the warp kernel | raw counter (KiB) | corrected bytes |
|---|---|---|
| read  (FETCH_SIZE, {nf} dispatches) | {f_k:.1f} | {rd:,.0f} |
| write (WRITE_SIZE, {nw} dispatches) | {w_k:.1f} | {wr:,.0f} |
| total | | {traffic:,.0f} |

Algorithmic bytes per launch (DESIGN.md section 5): {ALGO_BYTES:,} -> traffic / algorithmic = {traffic / ALGO_BYTES:.3f}
(the excess is the batch-shared table and inv_delta_C, which the algorithmic figure excludes).
""")
    # kernel stats of the wider rows (whole TPS++ module, whole recogniser, warp backward)
    for w, name in (("module", "module"), ("head", "recognizer"), ("backward", "warp_backward"),
                    ("module_bf16", "module_bf16"), ("module_x3", "module_bf16x3"), ("backbone_bf16", "backbone_bf16")):
        src = os.path.join(SRC, f"trace_{w}", f"{w}_kernel_stats.csv")
        if os.path.exists(src):
            rows = sorted(csv.DictReader(open(src)), key=lambda r: -float(r["TotalDurationNs"]))
            with open(os.path.join(DST, f"{TAG}_{name}_kernel_stats.csv"), "w", newline="") as f:
                wr_ = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
                wr_.writeheader()
                wr_.writerows(rows[:40])
            log = os.path.join(SRC, f"{w}.log")
            if os.path.exists(log):
                with open(os.path.join(DST, f"{TAG}_{name}_run.txt"), "w") as f:
                    f.write("".join(l for l in open(log) if "amdgpu.ids" not in l))
    cl = os.path.join(SRC, "conv_bf16.log")
    if os.path.exists(cl):
        with open(os.path.join(DST, f"{TAG}_conv_fp32_bf16_vs_miopen.txt"), "w") as f:
            f.write("".join(l for l in open(cl) if "amdgpu.ids" not in l))
    # round 6: every convolution call of the bf16 backbone with its shape / layouts / time, and the wide layers against the tiled kernel
    for logname, dst in (("backbone_layers.log", "backbone_bf16_layers.txt"), ("bench_wide.log", "conv3_wide_vs_tiled.txt")):
        pth = os.path.join(SRC, logname)
        if os.path.exists(pth):
            with open(os.path.join(DST, f"{TAG}_{dst}"), "w") as f:
                f.write("".join(l for l in open(pth) if "amdgpu.ids" not in l))
    mfma_summary()
    print(json.dumps(js, indent=1))


if __name__ == "__main__":
    main()
