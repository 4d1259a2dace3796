#!/usr/bin/env python3
"""Per run of tools/mode_pmc.sh: bench ms/step, and per smx kernel the mean dispatch duration (from the counter
records' own timestamps) and the mean of every counter; effective clock = GRBM_GUI_ACTIVE / 8 / duration
(MI355X_MICROARCH.md, DVFS give-back: rocprofv3 sums the 8 XCDs)."""
import collections, csv, glob, json, os, sys


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def main(o):
    runs = []
    for log in sorted(glob.glob(f"{o}/run*.log"), key=lambda p: int(os.path.basename(p)[3:-4])):
        tag = os.path.basename(log)[:-4]
        line = [l for l in open(log) if l.startswith("{")]
        b = json.loads(line[-1]) if line else {}
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{o}/{tag}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "smx::" not in r["Kernel_Name"]:
                    continue
                k = short(r["Kernel_Name"])
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                per[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        ks = {}
        for k, v in per.items():
            e = {c: sum(x) / len(x) for c, x in v.items()}
            e["_calls"] = len(v["_dur_us"])
            if "GRBM_GUI_ACTIVE" in e:
                e["clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8 / (e["_dur_us"] * 1e3)
            ks[k] = {a: round(b2, 4) if isinstance(b2, float) else b2 for a, b2 in e.items()}
        runs.append({"run": tag, "ms_per_step": b.get("ms_per_step"), "min_ms_per_step": b.get("min_ms_per_step"),
                     "launches": (b.get("roofline") or {}).get("launches"), "kernels": ks})
    print(json.dumps({"runs": runs}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
