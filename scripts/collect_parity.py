"""Folds gpurun_out/parity_gpu.jsonl (written by tests/conftest.py:record_parity during a
`pytest -m gpu` run on the GPU box) into the tracked profiles/parity_rNN.json: the last
record per (test, config), sorted, plus a per-test summary.

    python scripts/collect_parity.py [gpurun_out/parity_gpu.jsonl] [profiles/parity_r05.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> None:
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_gpu.jsonl")
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "parity_r05.json")
    last = {}
    with open(src) as f:
        for line in f:
            line = line.strip()
            if line:
                rec = json.loads(line)
                last[(rec["test"], rec["config"])] = rec
    recs = [last[k] for k in sorted(last)]
    # per test: the worst row against the oracle and - on the rows that differ from it by more
    # than 1e-4, which the float64 evaluation arbitrates - the worst GPU and the worst ORACLE
    # distance from float64 (a large first figure with a small second one means the oracle's own
    # float32 rounding is the far side)
    summary = {}
    for r in recs:
        s = summary.setdefault(r["test"], {"n_comparisons": 0})
        s["n_comparisons"] += 1
        for key, name in (("worst_row_err", "worst_row_err_vs_oracle"), ("worst_vs_float64", "worst_gpu_vs_float64"),
                          ("oracle_vs_float64", "worst_oracle_vs_float64"), ("n_rows_over_1e_4", "max_rows_over_1e-4_vs_oracle"),
                          ("worst_value_rel_err", "worst_value_rel_err"),
                          # round 4: float64 is the arbiter of every row (conftest.assert_float64_bar)
                          ("gpu_vs_f64_worst", "worst_gpu_vs_float64"), ("gpu_vs_f64_p999", "worst_gpu_p999_vs_float64"),
                          ("oracle_f32_vs_f64_worst", "worst_oracle_vs_float64"),
                          ("oracle_f32_vs_f64_p999", "worst_oracle_p999_vs_float64"),
                          ("gpu_vs_oracle_f32_worst", "worst_row_err_vs_oracle"),
                          # round 5: the operating-point tests (tests/_operating_point.py: summary)
                          ("fac_gpu_worst", "worst_gpu_vs_float64"), ("fac_orc_worst", "worst_oracle_vs_float64"),
                          ("sco_gpu_worst", "worst_gpu_own_item_scores_vs_float64"),
                          ("sco_orc_worst", "worst_oracle_own_item_scores_vs_float64"),
                          ("res_gpu_worst", "worst_gpu_backward_error"), ("res_orc_worst", "worst_oracle_backward_error"),
                          ("n_fac_gpu_over_1e_4", "max_rows_gpu_over_1e-4_vs_float64"),
                          ("n_fac_orc_over_1e_4", "max_rows_oracle_over_1e-4_vs_float64"),
                          ("n_rows_gpu_over_1e_4_vs_f64", "max_rows_gpu_over_1e-4_vs_float64"),
                          ("n_rows_oracle_over_1e_4_vs_f64", "max_rows_oracle_over_1e-4_vs_float64")):
            v = r.get(key)
            if v is not None:
                s[name] = max(s.get(name, 0), v)
        if "indices_bit_exact" in r:
            s["indices_bit_exact"] = bool(s.get("indices_bit_exact", True) and r["indices_bit_exact"])
    out = {"source": os.path.relpath(src, ROOT), "n_records": len(recs),
           "summary_by_test": summary, "records": recs}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{len(recs)} records -> {dst}")


if __name__ == "__main__":
    main()
