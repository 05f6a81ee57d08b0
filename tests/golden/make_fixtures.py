#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference tree (run in the build container only; the GPU box has
no /root/reference).  Everything written here is DATA the reference holds (inputs, committed .canz
outputs, token-id traces) or values derived from it — never reference source text.

  ref_canz/*.canz           committed reference containers (src/main.rs:594-611 layout pins)
  inputs/*                  the input files those containers were produced from (orig_hash16 / orig_len pins)
  tokens_*.u16              SmolLM2 token ids recovered from watchdog_decode_steps.jsonl ("sym" of every
                            decode_step record, in order; BOS not included)
  manifest.json             sizes, token counts, mismatch reports (off-by-one hazard)
"""
import json, os, shutil, sys
import numpy as np
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

def main():
    os.makedirs(f"{OUT}/ref_canz", exist_ok=True); os.makedirs(f"{OUT}/inputs", exist_ok=True)
    canz = {"asyoulik": "final_bench/cantrbry/asyoulik.canz", "fields": "final_bench/cantrbry/fields.canz",
            "alice29": "final_bench/cantrbry/alice29.canz", "enwik8_128kb_0": "final_bench/enwik8_samples/enwik8_128kb_0.canz"}
    inputs = {"asyoulik": "final_bench/cantrbry/asyoulik.txt", "fields": "final_bench/cantrbry/fields.c",
              "alice29": "final_bench/cantrbry/alice29.txt", "enwik8_128kb_0": "final_bench/enwik8_samples/enwik8_128kb_0"}
    man = {"canz": {}, "inputs": {}, "tokens": {}, "mismatch": {}}
    for k, p in canz.items():
        shutil.copyfile(f"{REF}/{p}", f"{OUT}/ref_canz/{k}.canz"); os.chmod(f"{OUT}/ref_canz/{k}.canz", 0o644)
        man["canz"][k] = {"bytes": os.path.getsize(f"{REF}/{p}"), "src": p}
    for k, p in inputs.items():
        dst = f"{OUT}/inputs/{os.path.basename(p)}"
        shutil.copyfile(f"{REF}/{p}", dst); os.chmod(dst, 0o644)
        man["inputs"][k] = {"bytes": os.path.getsize(f"{REF}/{p}"), "file": os.path.basename(p), "src": p}
    traces = {"asyoulik": "results_300s_nomem/asyoulik_selftest_smollm_20251003_121454",
              "alice29": "results_300s_nomem/alice29_selftest_smollm_20251003_043405",
              "enwik8_128kb_0": "results_300s_nomem/results_360s_nomem/enwik8_128kb_0_selftest_smollm_20251003_215646"}
    for k, d in traces.items():
        syms = []
        with open(f"{REF}/{d}/watchdog_decode_steps.jsonl") as f:
            for line in f:
                r = json.loads(line)
                if r.get("phase") == "decode_step":
                    assert r["i"] == len(syms); syms.append(r["sym"])
        mm = None
        if os.path.exists(f"{REF}/{d}/watchdog_mismatch.json"):
            mm = json.load(open(f"{REF}/{d}/watchdog_mismatch.json"))
            man["mismatch"][k] = mm
        valid = len(syms) if mm is None else mm["i"]          # decoded symbols are trustworthy only before the first mismatch
        a = np.array(syms[:valid], dtype=np.uint32); assert a.max() < 65536
        a.astype("<u2").tofile(f"{OUT}/tokens_{k}.u16")
        man["tokens"][k] = {"count": int(valid), "trace_records": len(syms), "src": f"{d}/watchdog_decode_steps.jsonl"}
        # baseline cross-entropy ledger (soft pin for real-weights runs)
        import csv
        with open(f"{REF}/{d}/proof.csv") as f:
            rows = list(csv.DictReader(f))
        man["tokens"][k]["xe_baseline_bits_sum"] = round(sum(float(r["cross_entropy_baseline_bits"]) for r in rows), 2)
        man["tokens"][k]["xe_windows"] = len(rows)
    json.dump(man, open(f"{OUT}/manifest.json", "w"), indent=1, sort_keys=True)
    print(json.dumps(man, indent=1))

if __name__ == "__main__":
    main()
