"""Summarise a `rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
--kernel-trace --output-format csv -d DIR -o p -- python bench.py ...` run per kernel: LDS bank-conflict share of the
LDS-active cycles and matrix-pipe busy share of the CU-busy cycles (4 SIMDs per CU).  Usage: pmc_summary.py DIR OUT.txt"""
import collections
import csv
import glob
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
f = glob.glob(src + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void jnr::", "").replace("jnr::", "")[:62]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(dst, "w") as out:
    out.write(f"# {' '.join(sys.argv)}\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
        idx, bc = v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0)
        mf, cu = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("SQ_BUSY_CU_CYCLES", 1)
        out.write(f"{k:62s} cu_busy {cu:14.0f}  lds_conflict/lds_active {bc / max(idx, 1):.2f}  mfma_busy/(4*cu_busy) {mf / max(cu, 1) / 4:.3f}\n")
