"""Per-kernel sums of every counter in a `rocprofv3 --pmc ... --output-format csv -d DIR` run, normalised per dispatch.
usage: pmc_dump.py DIR [name-filter]"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void jnr::", "").replace("jnr::", "")[:70] + " grid " + r.get("Grid_Size", "")
    if flt and flt not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = max(len(cnt[k]), 1)
    print(k, "dispatches", n)
    for c, x in sorted(v.items()): print(f"    {c:32s} {x / n:16.0f}")
