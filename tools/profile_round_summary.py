"""Summarise the passes of tools/profile_round.sh <tag> into profiles/<tag>_*.{csv,json} (per-kernel averages per launch):
    python tools/profile_round_summary.py r03"""
import collections
import csv
import glob
import json
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(R, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
sys.path.insert(0, R)


def short(n):
    return re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("void ", "").strip()[:80]


def counters(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for f in glob.glob(os.path.join(R, "gpurun_out", tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
    return agg, cnt


def main():
    # kernel stats
    st = glob.glob(os.path.join(R, "gpurun_out", TAG + "_stats", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        rows = list(csv.DictReader(open(st[0])))
        with open(os.path.join(OUT, TAG + "_a_kernel_stats_bench_cfg3.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "ms_per_step(4 steps incl. warm-up)"])
            for r in rows[:30]:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            "%.3f" % (float(r["TotalDurationNs"]) / 4e6)])
    fetch, nf = counters(TAG + "_fetch")
    write, nw = counters(TAG + "_write")
    sq, ns = counters(TAG + "_sq")
    with open(os.path.join(OUT, TAG + "_b_pmc_per_kernel.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "avg_FETCH_SIZE_KB_per_launch", "avg_WRITE_SIZE_KB_per_launch", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
                    "SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"])
        for k in sorted(fetch, key=lambda k: -fetch[k]["FETCH_SIZE"]):
            n = max(nf[k], 1)
            s = sq.get(k, {})
            w.writerow([k, nf[k], "%.1f" % (fetch[k]["FETCH_SIZE"] / n), "%.1f" % (write.get(k, {}).get("WRITE_SIZE", 0.0) / max(nw[k], 1))] +
                       ["%.4g" % s.get(c, 0.0) for c in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU",
                                                        "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")])
    dom = sorted((k for k in fetch if "k_conv3_f16x3" in k), key=lambda k: -nf[k])
    if dom:
        k = dom[0]
        fk, wk = fetch[k]["FETCH_SIZE"] / max(nf[k], 1), write[k]["WRITE_SIZE"] / max(nw[k], 1)
        tj = {"kernel": k, "plc_mode": "f16x3", "fused": ("<2" in k or "ILi2E" in k), "launches": nf[k],
              "avg_fetch_KB_per_launch": fk, "avg_write_KB_per_launch": wk,
              "traffic_bytes_per_launch": (fk + wk) * 1024.0,
              "traffic_bytes_per_launch_if_every_fetch_were_16B_per_lane": (2.0 * fk + wk) * 1024.0,
              "correction": "lower bound: none applied; upper bound: FETCH_SIZE doubled (the MI355X guide: gfx950 tallies 16-B-per-lane streaming reads at half their bytes).  Fused pair (<2>): the only HBM input is the 3-channel parent (4 B/lane loads); the "
                            "weight-fragment stream (2.4 MB per plane, L2-resident) is 16 B/lane, where FETCH_SIZE reads half "
                            "(MI355X guide), so the L2->CU weight traffic is about twice its share of this figure",
              "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 "
                        "--warmup 1 --train-steps 0 --no-cpu-baseline --no-hbm-kernels; profiles/%s_b_pmc_per_kernel.csv" % TAG}
        lk = [k2 for k2 in fetch if "k_lift_fused_f16" in k2]
        if lk:
            k2 = lk[0]
            steps_pmc = 2.0                                    # the PMC passes run --steps 1 --warmup 1
            tj["lifting"] = {"kernel": k2, "launches_per_forward": nf[k2] / steps_pmc,
                             "fetch_bytes_per_forward": fetch[k2]["FETCH_SIZE"] * 1024.0 / steps_pmc,
                             "write_bytes_per_forward": write[k2]["WRITE_SIZE"] * 1024.0 / steps_pmc,
                             "traffic_bytes_per_forward": (fetch[k2]["FETCH_SIZE"] + write[k2]["WRITE_SIZE"]) * 1024.0 / steps_pmc,
                             "note": "sum over the launches of ONE lldwt_lifting_forward call (4 B/lane loads: no FETCH_SIZE correction)"}
        import bench
        tj["source_hashes"] = bench.source_hashes()      # bench.py drops the figures when a kernel source changes
        tj["csv"] = "profiles/%s_b_pmc_per_kernel.csv" % TAG
        json.dump(tj, open(os.path.join(OUT, "traffic_current.json"), "w"), indent=1)
        json.dump(tj, open(os.path.join(OUT, TAG + "_traffic.json"), "w"), indent=1)
        print(json.dumps(tj))
    # does a kernel's launch count scale with the number of steps?  (4 steps vs 8 steps incl. the warm-up)
    st8 = glob.glob(os.path.join(R, "gpurun_out", TAG + "_stats8", "**", "*kernel_stats.csv"), recursive=True)
    if st and st8:
        c4 = {short(r["Name"]): int(r["Calls"]) for r in csv.DictReader(open(st[0]))}
        c8 = {short(r["Name"]): int(r["Calls"]) for r in csv.DictReader(open(st8[0]))}
        rows = [{"kernel": k, "calls_4_steps": c4[k], "calls_8_steps": c8.get(k), "per_step": (c8.get(k, 0) - c4[k]) / 4.0,
                 "setup": c4[k] - (c8.get(k, 0) - c4[k])} for k in sorted(c4, key=lambda k: -c4[k])[:40]]
        json.dump(rows, open(os.path.join(OUT, TAG + "_c_calls_vs_steps.json"), "w"), indent=1)
        for r_ in rows:
            if "copyBuffer" in r_["kernel"] or "fillBuffer" in r_["kernel"]:
                print(r_)


if __name__ == "__main__":
    main()
