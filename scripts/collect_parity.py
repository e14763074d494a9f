"""Folds gpurun_out/parity_gpu.jsonl (written by tests/conftest.py:record_parity during a
`pytest -m gpu` run on the GPU box) into the tracked profiles/parity_rNN.json: the last
record per (test, config), sorted, plus the worst figure per test.

    python scripts/collect_parity.py [gpurun_out/parity_gpu.jsonl] [profiles/parity_r03.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> None:
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_gpu.jsonl")
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "parity_r03.json")
    last = {}
    with open(src) as f:
        for line in f:
            line = line.strip()
            if line:
                rec = json.loads(line)
                last[(rec["test"], rec["config"])] = rec
    recs = [last[k] for k in sorted(last)]
    worst = {}
    for r in recs:
        e = r.get("worst_row_err")
        if e is not None:
            worst[r["test"]] = max(worst.get(r["test"], 0.0), e)
    out = {"source": os.path.relpath(src, ROOT), "n_records": len(recs),
           "worst_row_err_by_test": worst, "records": recs}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{len(recs)} records -> {dst}")


if __name__ == "__main__":
    main()
