#!/usr/bin/env python3
"""Summarises the rocprofv3 --pmc passes of tools/profile_round.sh: HBM traffic per launch of the dominant
kernel of each workload (FETCH_SIZE / WRITE_SIZE with the gfx950 corrections of MI355X_MICROARCH.md: a wide
coalesced read stream is tallied at half its bytes; writes as they are) and the SQ counters used to steer
the optimisation.  usage: profile_summary.py <raw dir> <summary dir>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from resampler_amd import provenance

raw, out = sys.argv[1], sys.argv[2]
KERNELS = {"fir": ("fir_split", "fir_periodic"), "fft": ("fft_ola",), "c4": ("fir_lockstep",), "c5": ("fir_split", "fir_periodic")}
BENCH = {"fir": "bench_n1.json", "fft": "bench_fft.json", "c4": "bench_c4.json", "c5": "bench_c5.json"}
latest = {}
for w, names in KERNELS.items():
    tot = collections.defaultdict(list)
    for f in glob.glob(f"{raw}/pmc_{w}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(n in r["Kernel_Name"] for n in names):
                tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
    mean = {k: sum(v) / len(v) for k, v in tot.items()}
    try:
        bench = json.loads(open(os.path.join(out, BENCH[w])).read().strip().splitlines()[-1])
    except Exception:
        bench = None
    roof = bench["roofline"] if bench else {}
    alg = roof.get("algorithmic_bytes")
    kk = (bench or {}).get("config", {}).get("steps_per_launch", 1) if w == "c4" else 1
    if w == "c4" and kk > 1:
        # a run of kk lock-step calls is several launches (planner kernels, one bulk kernel per rate pair, repair, tail
        # copy): the traffic of ALL of them per run, against the algorithmic bytes of the run's kk steps
        run_names = ("fir_lockstep", "fir_split", "split_items", "fir_repair", "fir_tail_copy", "fir_periodic")
        sums, runs = collections.defaultdict(float), collections.defaultdict(int)
        for f in glob.glob(f"{raw}/pmc_{w}_*/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if any(n in r["Kernel_Name"] for n in run_names):
                    sums[r["Counter_Name"]] += float(r["Counter_Value"])
                    # (a run = one repair launch behind its bulk kernels; the chain kernel is launched once more than runs are
                    # computed -- the last run planned ahead is never asked for --, which read 6 runs as 7: round 6's first pass)
                    if "fir_repair" in r["Kernel_Name"]:
                        runs[r["Counter_Name"]] += 1
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            if runs[c]:
                mean[c] = sums[c] / runs[c]
                tot[c] = [0.0] * runs[c]
        alg = alg * kk if alg else alg
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        fetch = mean["FETCH_SIZE"] * 1024 * 2      # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
        write = mean["WRITE_SIZE"] * 1024
        ent = {"kernel": roof.get("kernel"), "command": "python3 bench.py (see tools/profile_round.sh PCMD[%s]); one rocprofv3 --pmc pass per counter" % w,
               "raw": {"FETCH_SIZE": {"dispatches": len(tot["FETCH_SIZE"]), "mean_kb": mean["FETCH_SIZE"]},
                       "WRITE_SIZE": {"dispatches": len(tot["WRITE_SIZE"]), "mean_kb": mean["WRITE_SIZE"]}},
               "correction": "FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md: a coalesced stream is tallied at 1/2); WRITE_SIZE as is"
                             + (" (fft_pair.hip stores whole 8-byte frames, non-temporal; every run after a stream's first reads one halo block)" if w == "fft" else ""),
               "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
               "algorithmic_bytes_per_launch": alg,
               "ratio_traffic_to_algorithmic": (fetch + write) / alg if alg else None}
        json.dump(ent, open(os.path.join(out, f"traffic_{w}.json"), "w"), indent=1)
        latest[w] = {"hbm_bytes_per_launch": int((fetch + write) / kk), "kernel": roof.get("kernel"),
                     "source": f"profiles/<tag>/traffic_{w}.json",
                     # the kernel sources the counters belong to: bench.py reports the traffic only while it runs these
                     "kernel_sources_sha": provenance.kernel_sources_sha(w)}
        if kk > 1:   # (bench.py's config-4 line is per 512-frame step: the run's traffic / its steps)
            latest[w]["steps_per_launch"] = kk
            latest[w]["hbm_bytes_per_run"] = int(fetch + write)
    with open(os.path.join(out, f"pmc_{w}.txt"), "w") as f:
        f.write("%s, per-dispatch means (rocprofv3 --pmc, separate passes)\n" % (roof.get("kernel") or w))
        for k in sorted(mean):
            f.write("%-28s %18.1f   n=%d\n" % (k, mean[k], len(tot[k])))
        if "GRBM_GUI_ACTIVE" in mean:
            cyc = mean["GRBM_GUI_ACTIVE"] / 8
            f.write("cycles per dispatch (GRBM_GUI_ACTIVE / 8 XCDs): %.0f\n" % cyc)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mean:
                f.write("matrix pipe busy: %.1f %% of 1024 SIMDs x cycles\n" % (100 * mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)))
        if "SQ_WAVE_CYCLES" in mean and "SQ_WAIT_ANY" in mean:
            f.write("wave time waiting at s_waitcnt: %.1f %%, issue stalls: %.1f %%, issuing: %.1f %%\n" % (
                100 * mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"], 100 * mean.get("SQ_WAIT_INST_ANY", 0) / mean["SQ_WAVE_CYCLES"],
                100 * mean.get("SQ_ACTIVE_INST_ANY", 0) / mean["SQ_WAVE_CYCLES"]))
    print(open(os.path.join(out, f"pmc_{w}.txt")).read())
if "fir" in latest and os.path.exists(os.path.join(out, "bench_n1.json")):
    b = json.loads(open(os.path.join(out, "bench_n1.json")).read().strip().splitlines()[-1])
    latest["fir"]["variant_name"] = b["roofline"]["kernel"]
json.dump(latest, open(os.path.join(out, "traffic_latest.json"), "w"), indent=1)
print(json.dumps(latest))
