#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs of the conv stack into HBM bytes per pass.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <glimpse steps in the run> <batch> [train|eval|backward] [json to update] [name of the summary file for the json]
(train: counters of a `--mode train` run — only the FORWARD conv-stack kernels are summed: the data-gradient launches
of pw_mfma_kernel carry `true` as their third template argument and are left out; backward: the same run, only the
conv-stack BACKWARD kernels — BatchNorm-backward reductions, fused / unfused data and weight gradients, stem weight
gradient, pooling / upsample / copy backward — per glimpse step)

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KB;
FETCH_SIZE reports half of the bytes of wide coalesced streaming reads, so read bytes = 2 * FETCH_SIZE * 1024.
"""
import collections
import csv
import re
import sys

CONV = ("stem_mfma_kernel", "dw3x3", "dwpw_eval_kernel", "pw_mfma_kernel", "pw_narrow_kernel", "pw_dir_kernel", "pw_res_kernel", "pw_xs_kernel", "pw_x3_kernel", "w_split3_kernel", "addact_kernel",
        "spp_kernel", "spp4_kernel", "upsample_kernel", "conv3_mfma", "bn_finalize_all_kernel")


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("jnr::", "")
    return re.sub(r"\(.*", "", name)


TRAIN = False
BACKWARD = False
BWD = ("bn_bwd_", "pw_bwd_", "dw_bwd_", "stem_bwd_", "spp_bwd", "spp4_bwd", "upsample_bwd", "grad_copy", "wpart_reduce", "conv3_bwd_", "w_split3_t")


def is_data_gradient(k):
    # (pw_x3_kernel<.., SLOTS = true>: the wide layers' data gradient over the step slots, round 4)
    return ((k.startswith("pw_mfma_kernel") and ", true," in k) or bool(re.match(r"pw_(dir|res)_kernel<\d+(, \d+, \d+, \d+)?, true", k))
            or (k.startswith("pw_x3_kernel") and k.rstrip().endswith(", true>")))


def load(path, counter):
    tot, calls = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if BACKWARD:
            if not (any(c in k for c in BWD) or is_data_gradient(k)):
                continue
            tot[k] += float(r["Counter_Value"])
            calls[k] += 1
            continue
        if not any(c in k for c in CONV + (("bn_finalize_kernel",) if TRAIN else ())):
            continue
        if TRAIN and k.startswith("pw_mfma_kernel") and ", true," in k:
            continue
        if TRAIN and is_data_gradient(k):      # WT = true / SLOTS = true: data gradients
            continue
        tot[k] += float(r["Counter_Value"])
        calls[k] += 1
    return tot, calls


def main():
    global TRAIN, BACKWARD
    TRAIN = len(sys.argv) > 5 and sys.argv[5] == "train"
    BACKWARD = len(sys.argv) > 5 and sys.argv[5] == "backward"
    fetch, calls = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    steps, batch = int(sys.argv[3]), int(sys.argv[4])
    print(f"{'kernel':58s} {'calls':>6s} {'FETCH_SIZE_KB':>14s} {'WRITE_SIZE_KB':>14s}")
    for k in sorted(fetch, key=lambda k: -(fetch[k] + write[k])):
        print(f"{k:58s} {calls[k]:6d} {fetch[k]:14.0f} {write[k]:14.0f}")
    f, w = sum(fetch.values()), sum(write.values())
    rd, wr = 2 * f * 1024 / steps, w * 1024 / steps
    algo = (2 * 17.44e6 if BACKWARD else 17.44e6) * 4 * batch
    what = "conv-stack backward" if BACKWARD else "conv stack"
    print(f"# {what}, {steps} glimpse steps: FETCH {f:.0f} KB, WRITE {w:.0f} KB")
    print(f"# per glimpse step ({'backward of ' if BACKWARD else ''}one pass over {batch} patches): reads 2*{f * 1024 / steps / 1e9:.3f} = {rd / 1e9:.3f} GB, "
          f"writes {wr / 1e9:.3f} GB, total {(rd + wr) / 1e9:.3f} GB")
    print(f"# algorithmic ({'2 (in + out) = 2 * ' if BACKWARD else 'SURVEY 8d, fp32: '}17.44 M elems * 4 B * {batch}) = {algo / 1e9:.3f} GB -> traffic / algorithmic = "
          f"{(rd + wr) / algo:.2f}")
    if len(sys.argv) > 6:                  # bench.py reads the per-pass bytes of the latest committed PMC passes from here
        import json, os
        path = sys.argv[6]
        cur = json.load(open(path)) if os.path.exists(path) else {}
        key = "backward" if BACKWARD else "train" if TRAIN else "rollout"
        cur[key] = rd + wr
        if len(sys.argv) > 7:
            cur.setdefault("files", {})[key] = sys.argv[7]
        json.dump(cur, open(path, "w"))


if __name__ == "__main__":
    main()
