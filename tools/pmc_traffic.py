#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md
prescribes) of `bench.py` into profiles/<tag>_pmc_traffic.json: mean counter value per launch of the
row-/column-phase kernels and the HBM-side bytes per mini-batch.

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; WRITE_SIZE is exact for
16-B-per-lane stores; FETCH_SIZE reports half of the bytes of wide (16 B per lane) reads on gfx950 and
is doubled here (the kernels' parameter-row and A-row reads are 16 B per lane; the narrower CSR reads
are a few percent of the total and are doubled with them -- an upper bound).

usage: tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv workload batch tag
"""
import collections
import csv
import json
import sys


def means(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"]
            key = None
            for pat, k_ in (("k_row_phase", "row_phase"), ("k_ffm_row_phase", "row_phase"), ("k_col_phase", "col_phase"), ("k_col_sparse", "col_phase"),
                            ("k_ffm_col_phase", "col_phase"), ("k_ffm_refresh", "refresh"), ("k_heavy_partial", "heavy_partial"), ("k_heavy_apply", "heavy_apply"),
                            ("k_ffm_heavy_partial", "heavy_partial"), ("k_ffm_heavy_apply", "heavy_apply"),
                            ("k_singles", "singles"), ("k_psgd_", "psgd_step"), ("k_prox_", "psgd_step"),
                            ("k_fm_predict", "predict"), ("k_ffm_predict", "predict"), ("k_interleave_orders", "predict_interleave")):
                if pat in name:
                    key = k_
                    break
            if key:
                acc[key].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch_csv, write_csv, workload, batch, tag = sys.argv[1:6]
    f, nf = means(fetch_csv, "FETCH_SIZE")
    w, nw = means(write_csv, "WRITE_SIZE")
    per_kernel = {k: {"FETCH_SIZE_KiB_raw": f[k], "FETCH_bytes_corrected": 2 * f[k] * 1024, "WRITE_SIZE_KiB": w.get(k, 0.0),
                      "WRITE_bytes": w.get(k, 0.0) * 1024, "launches_sampled": nf[k]} for k in f}
    # one mini-batch = one launch of the row and of the column phase; the heavy kernels run once per batch that has
    # heavy features -- weight every family by its launches relative to the row phase
    base = max(nf.get("row_phase", 1), 1)
    # (decisionFunction's kernels are reported per launch = per pass over the shard; they are not part of a mini-batch)
    total = sum((v["FETCH_bytes_corrected"] + v["WRITE_bytes"]) * nf[k] / base for k, v in per_kernel.items() if not k.startswith("predict"))
    out = {"workload": workload, "batch": int(batch), "per_kernel": per_kernel, "hbm_bytes_per_minibatch": total,
           "note": "FETCH_SIZE doubled (gfx950, 16 B/lane reads), WRITE_SIZE exact; separate --pmc passes"}
    path = "profiles/%s_pmc_traffic.json" % tag
    json.dump(out, open(path, "w"), indent=1)
    print(path, "bytes per mini-batch: %.1f MB" % (total / 1e6))


if __name__ == "__main__":
    main()
