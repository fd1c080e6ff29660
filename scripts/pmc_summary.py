#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*) -> JSON for profiles/.

    python scripts/pmc_summary.py OUT.json name=counter_collection.csv [name=...] [workload=bench_line.json]

workload=: the bench.py line of one of the passes; its genome size, read length and pairs per launch are recorded so that bench.py
only quotes these figures for the same workload.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced streaming read (128-byte requests tallied as 64) and says other access widths are uncalibrated.  Calibrated here
(scripts/probes/fetch_calib.hip, profiles/r5_fetch_calib.log; 8 GB buffer, 64 M accesses per pattern): random 64-byte records read whole by four lanes,
one 16-byte load per random line and random 8-byte words all give FETCH_SIZE = 1.000 x the 64-byte lines touched; only the streaming read gives 0.500 x.
The kernels of the select path read scattered records, so hbm_read_bytes = FETCH_SIZE as reported (every line once); the doubled figure the summaries of
rounds 1-4 carried is kept as hbm_read_bytes_per_launch_if_streaming (it is the right one only for kernels that stream: BCL staging, sorts, BAM encode,
deflate)."""
import collections, csv, json, re, sys

def main():
    out = sys.argv[1]
    kernels = collections.defaultdict(lambda: {"launches": 0})
    workload = None
    gapped_order = {}
    for arg in sys.argv[2:]:
        name, path = arg.split("=", 1)
        if name == "workload":
            line = [l for l in open(path).read().splitlines() if l.startswith("{")][-1]
            cfg = json.loads(line)["config"]
            workload = {"genome_bases": cfg["genome_bases"], "read_length": cfg["read_length"], "pairs_per_launch": cfg["pairs_per_step"], "index_entries": cfg["index_entries"]}
            continue
        seen = set()
        for r in csv.DictReader(open(path)):
            m = re.search(r"(k_\w+)", r["Kernel_Name"])
            if not m:
                continue
            names = [m.group(1)]
            if m.group(1) == "k_gapped_jobs":
                # launched twice per select call, for the fragment stage and then for the mate rescue: the dispatches alternate
                order = gapped_order.setdefault(path, {})
                stage = order.setdefault(r["Dispatch_Id"], len(order) % 2)
                names.append("k_gapped_jobs:" + ("rescue" if stage else "fragments"))
            for name in names:
                k = kernels[name]
                k[r["Counter_Name"]] = k.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                key = (path, name, r["Dispatch_Id"])
                if key not in seen:
                    seen.add(key)
                    k.setdefault("_launches_" + path, 0)
                    k["_launches_" + path] += 1
    res = {}
    for name, k in sorted(kernels.items()):
        launches = max(v for kk, v in k.items() if kk.startswith("_launches_"))
        e = {"launches": launches}
        for c, v in k.items():
            if c.startswith("_") or c == "launches":
                continue
            e[c] = v
        if "FETCH_SIZE" in e:
            e["hbm_read_bytes_per_launch"] = e["FETCH_SIZE"] * 1024 / launches
            e["hbm_read_bytes_per_launch_if_streaming"] = 2.0 * e["FETCH_SIZE"] * 1024 / launches
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE"] * 1024 / launches
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
        if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
            wc = e["SQ_WAVE_CYCLES"]
            e["wait_any_frac"] = e.get("SQ_WAIT_ANY", 0) / wc
            e["active_inst_frac"] = e.get("SQ_ACTIVE_INST_ANY", 0) / wc
            if e.get("SQ_ACTIVE_INST_VALU"):
                e["valu_lane_utilisation"] = e.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * e["SQ_ACTIVE_INST_VALU"])
        if e.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"]
        res[name] = e
    if workload:
        res["workload"] = workload
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out, "kernels:", ", ".join(res))

if __name__ == "__main__":
    main()
