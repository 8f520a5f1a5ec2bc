#!/usr/bin/env python3
"""Per-kernel summary of three rocprofv3 --pmc passes (SQ/GRBM set, FETCH_SIZE, WRITE_SIZE) -> JSON.

usage: pmc_profile_summary.py DIR_SQ DIR_FETCH DIR_WRITE OUT.json
Definitions (MI355X_MICROARCH.md): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8);
clock = GRBM_GUI_ACTIVE / 8 / duration; HBM-side bytes = (2 * FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE reads half of a
wide coalesced stream).
Short dispatches: GRBM_GUI_ACTIVE spans more than the kernel on dispatches under ~30 us (the quotient GRBM_GUI_ACTIVE / 8 / duration
reads 3-16 "GHz" there; the guide notes it reads high below ~0.3 ms), so for kernels under 30 us - or whose derived clock exceeds
2.45 GHz, above what the chip can run - the active-cycle denominator is duration x the reference clock of the same pass (the
duration-weighted clock of its kernels of >= 50 us); "normalised_by" says which form an entry uses."""
import collections, csv, glob, json, re, sys


def short(name):
    m = re.search(r'gemm_(x6|f32)_kernel<(.*?)(?:paths_epi::)?(Epi\w+)', name)
    if m:
        args = [t.strip() for t in m.group(2).split(",")]
        if m.group(1) == "x6":          # <planes, WTM, WTN, PF, ADD, Epi>
            return f"gemm_{'h3' if args[0] == '2' else 'x6'}<{args[1]},{args[2]}>{m.group(3)}"
        return f"gemm_f32<{args[0]},{args[1]}>{m.group(3)}"
    m = re.search(r'::(\w+_kernel)(<\d+>)?\(', name)
    if m:
        return m.group(1) + (m.group(2) or "")
    m = re.search(r'::(\w+)\(', name)
    return m.group(1) if m else name[:40]


def load(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[k]['_dur'].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        agg[k]['_ids'].append(float(r['Dispatch_Id']))
    return {k: {c: sum(v) / len(v) for c, v in d.items() if c != '_ids'} | {'_n': len(set(d['_ids']))} for k, d in agg.items()}


sq, fe, wr = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
out = {"commands": ["rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                    "SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline",
                    "same with --pmc FETCH_SIZE, and with --pmc WRITE_SIZE (separate passes)"],
       "definitions": {"mfma_util": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8)", "clock_ghz": "GRBM_GUI_ACTIVE/8 / duration",
                       "traffic_bytes_per_launch": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH correction)",
                       "valu_issue_frac / wait_frac": "SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, SQ_WAIT_ANY / SQ_WAVE_CYCLES"},
       "kernels": {}}
long_k = [(m['_dur'] * m['_n'], m.get('GRBM_GUI_ACTIVE', 0) / 8 / m['_dur'] / 1e3) for m in sq.values() if m['_dur'] >= 50 and m.get('GRBM_GUI_ACTIVE', 0)]
ref_clock = sum(w * c for w, c in long_k) / sum(w for w, _ in long_k) if long_k else 2.0
out["reference_clock_ghz"] = round(ref_clock, 3)
out["definitions"]["short kernels"] = ("avg_us < 30 or derived clock > 2.45 GHz: active cycles = duration x reference_clock_ghz (the duration-weighted "
                                        "GRBM clock of this pass's kernels >= 50 us) instead of GRBM_GUI_ACTIVE / 8; see normalised_by")
for k, m in sorted(sq.items(), key=lambda kv: -kv[1]['_dur'] * kv[1]['_n']):
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8
    grbm_clock = cyc / m['_dur'] / 1e3 if m['_dur'] else None
    by = "GRBM_GUI_ACTIVE"
    if m['_dur'] and (m['_dur'] < 30 or (grbm_clock or 0) > 2.45 or not cyc):
        cyc, by = m['_dur'] * 1e3 * ref_clock, "duration x reference clock"
    e = {"dispatches": m['_n'], "avg_us": round(m['_dur'], 1), "clock_ghz": round(grbm_clock, 2) if grbm_clock else None,
         "normalised_by": by,
         "mfma_util": round(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc), 3) if cyc else None}
    wc = m.get('SQ_WAVE_CYCLES', 0)
    if wc:
        e["valu_issue_frac"] = round(m.get('SQ_ACTIVE_INST_VALU', 0) / wc, 3)
        e["wait_frac"] = round(m.get('SQ_WAIT_ANY', 0) / wc, 3)
        e["waves_per_simd_avg"] = round(wc * 4 / (1024 * cyc), 2) if cyc else None
    e["lds_bank_conflict_cycles"] = m.get('SQ_LDS_BANK_CONFLICT', 0.0)
    if k in fe and k in wr:
        f_kb, w_kb = fe[k].get('FETCH_SIZE', 0.0), wr[k].get('WRITE_SIZE', 0.0)
        e["FETCH_SIZE_KB_raw"], e["WRITE_SIZE_KB"] = round(f_kb, 1), round(w_kb, 1)
        e["traffic_bytes_per_launch"] = int((2 * f_kb + w_kb) * 1024)
        e["hbm_gbps"] = round(e["traffic_bytes_per_launch"] / m['_dur'] / 1e3, 1) if m['_dur'] else None
    out["kernels"][k] = e
# alias read by bench.py: the output-gate GEMM
for k in out["kernels"]:
    if "EpiLstmO" in k:                # (EpiLstmO_: the raw pre-activation form of the inference path)
        out["kernels"].setdefault("EpiLstmO", out["kernels"][k])
        break
json.dump(out, open(sys.argv[4], "w"), indent=1)
print("wrote", sys.argv[4], "kernels:", len(out["kernels"]))
