#!/usr/bin/env python3
"""Condense a tools/collect_profile.sh output directory into one JSON (committed under profiles/)."""
import collections, csv, glob, json, sys

def short(name):
    """'void smx::(anonymous namespace)::k_ln_bwd<4, 1>(float*, ...)' -> 'smx::k_ln_bwd<4, 1>'"""
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0]


def csrc_sha256():
    """Same value as bench.py's csrc_sha256(): the kernel sources this profile belongs to."""
    import hashlib, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    c = os.path.join(root, "tensor-cuda-fft-_amd", "csrc")
    files = sorted(glob.glob(os.path.join(c, "*.hip")) + glob.glob(os.path.join(c, "*.h")) +
                   [os.path.join(c, "build.sh"), os.path.join(root, "include", "smx.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def main(o, tag, cmd="python3 bench.py --no-cpu-baseline --steps 50 --warmup 10", git_sha="unknown",
         lib_sha=None):
    out = {"tag": tag, "command": cmd, "git_sha": git_sha, "libsmx_sha256": lib_sha, "csrc_sha256": csrc_sha256(),
           "notes": "FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts half the bytes of a "
                    "coalesced stream (MI355X_MICROARCH.md, HBM) -> read bytes = 2*FETCH_SIZE*1024; "
                    "check: 2*FETCH of the forward launch = x (268.4 MB) + tables, WRITE = y + saved spectrum"}
    ks = glob.glob(f"{o}/stats/*/*kernel_stats.csv")
    if ks:
        rows = [r for r in csv.DictReader(open(ks[0])) if "smx::" in r["Name"]]
        out["kernel_stats"] = [{"name": short(r["Name"]), "calls": int(r["Calls"]),
                                "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows]
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(f"{o}/pmc_*/")):
        for f in glob.glob(f"{d}*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "smx::" in r["Kernel_Name"]:
                    pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if "End_Timestamp" in r:     # (the dispatch's own duration under the counter pass)
                        pmc[short(r["Kernel_Name"])]["_dur_us_under_pmc"].append(
                            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out["counters_per_launch"] = {}
    for k, v in pmc.items():
        e = {c: sum(x) / len(x) for c, x in v.items()}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_read_bytes"] = 2 * e["FETCH_SIZE"] * 1024
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
            e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        out["counters_per_launch"][k] = e
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main(*sys.argv[1:6])
