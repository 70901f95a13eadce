#!/usr/bin/env python3
"""KAT ingestion for the reference's two vector formats, so that the
moment the files its build.sh downloads are available every precompile can be checked:

  * matter-labs / geth-fuzzer CSV (reference src/test.c:63-72): header line, then `input_hex,output_hex`
  * go-ethereum JSON (reference go/blst_eip2537_test.go:18-29): [{Input, Expected, Name, Gas, NoBenchmark}]
    `fail-*.json` files carry {Input, ExpectedError, Name}: any error is accepted there, like the
    reference's Go/Rust harnesses (go/blst_eip2537_test.go:79-84, rust/src/lib.rs:366-371); the
    C harness's per-file expected codes (src/test.c:144-165, 481-511, 564-585) are applied to the
    CSV failure files.

    python tools/run_kat.py <file-or-dir> [...]        # uses the product library through its C-ABI

File name -> precompile follows the reference's own mapping (build.sh:17-49).
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CSV_FILES = {   # file stem -> (op, expected error code or None for success files)
    "g1_add": ("g1_add", None), "g1_mul": ("g1_mul", None), "g1_multiexp": ("g1_multiexp", None),
    "g2_add": ("g2_add", None), "g2_mul": ("g2_mul", None), "g2_multiexp": ("g2_multiexp", None),
    "pairing": ("pairing", None), "fp_to_g1": ("map_fp_to_g1", None), "fp2_to_g2": ("map_fp2_to_g2", None),
    # the C harness feeds these two files to the MUL precompiles (160- / 288-byte rows: src/test.c:144-165, 337-358)
    "g1_not_on_curve": ("g1_mul", 1), "g2_not_on_curve": ("g2_mul", 1),
    "invalid_subgroup_for_pairing": ("pairing", 2),
    "invalid_fp_encoding": ("map_fp_to_g1", 3), "invalid_fp2_encoding": ("map_fp2_to_g2", 3),
}
JSON_FILES = {
    "blsG1Add": "g1_add", "blsG1Mul": "g1_mul", "blsG1MultiExp": "g1_multiexp",
    "blsG2Add": "g2_add", "blsG2Mul": "g2_mul", "blsG2MultiExp": "g2_multiexp",
    "blsPairing": "pairing", "blsMapG1": "map_fp_to_g1", "blsMapG2": "map_fp2_to_g2",
}


def _hex(s):
    s = s.strip()
    return bytes.fromhex(s[2:] if s.startswith("0x") else s)


def load(path):
    """-> list of (name, op, input, expected_bytes|None, expected_code|None|'any')"""
    stem = os.path.splitext(os.path.basename(path))[0]
    out = []
    if path.endswith(".csv"):
        op, code = CSV_FILES[stem]
        with open(path, newline="") as f:
            rows = list(csv.reader(f))[1:]                 # header row skipped (src/test.c:63)
        for i, row in enumerate(rows):
            if not row:
                continue
            inp = _hex(row[0])
            if code is None:
                out.append(("%s[%d]" % (stem, i), op, inp, _hex(row[1]), None))
            else:
                out.append(("%s[%d]" % (stem, i), op, inp, None, code))
    else:
        fail = stem.startswith("fail-")
        op = JSON_FILES[stem[5:] if fail else stem]
        with open(path) as f:
            for v in json.load(f):
                if fail or "ExpectedError" in v:
                    out.append((v.get("Name", stem), op, _hex(v["Input"]), None, "any"))
                else:
                    out.append((v.get("Name", stem), op, _hex(v["Input"]), _hex(v["Expected"]), None))
    return out


# the reference's C harness runs every multiexp row through the Bos-Coster entry point too
# (src/test.c:208-228, 401-421); the naive one is exported by the same header (src/eip2537.h:47,55)
ALSO = {"g1_multiexp": ["g1_multiexp_bc", "g1_multiexp_naive"], "g2_multiexp": ["g2_multiexp_bc", "g2_multiexp_naive"]}


def run(cases, executor=None, ops=None):
    """Returns (n_ok, failures).  `executor` defaults to the product's Eip2537Executor; `ops` restricts
    the run to those operations (the multiexp / pairing ones need the GPU)."""
    if executor is None:
        from blst_eip2537_amd import Eip2537Executor as executor
    from blst_eip2537_amd import Eip2537Error
    ok, failures = 0, []
    for name, op, inp, want, code in cases:
        if ops is not None and op not in ops:
            continue
        for fn in [op] + ALSO.get(op, []):
            try:
                got, gcode = getattr(executor, fn)(inp), 0
            except Eip2537Error as e:
                got, gcode = None, e.code
            good = (got == want) if want is not None else (gcode != 0 if code == "any" else gcode == code)
            if good:
                ok += 1
            else:
                failures.append((name, fn, gcode))
    return ok, failures


if __name__ == "__main__":
    paths = []
    for a in sys.argv[1:]:
        if os.path.isdir(a):
            paths += [os.path.join(a, f) for f in sorted(os.listdir(a)) if f.endswith((".csv", ".json"))]
        else:
            paths.append(a)
    total_bad = 0
    for p in paths:
        stem = os.path.splitext(os.path.basename(p))[0]
        if stem not in CSV_FILES and stem not in JSON_FILES and stem[5:] not in JSON_FILES:
            print("skip", p)
            continue
        ok, bad = run(load(p))
        total_bad += len(bad)
        print("%-40s %4d ok %3d FAILED %s" % (os.path.basename(p), ok, len(bad), bad[:3] if bad else ""))
    sys.exit(1 if total_bad else 0)
