"""Derived per-kernel figures from the two SQ PMC passes of tools/round_measurements.sh (appended to profiles/*_pmc_conv_kernels.txt).
mfma_pipe = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles) with kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs;
lds_busy = SQ_LDS_IDX_ACTIVE / (256 CUs x kernel cycles); lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;
waiting / stalled / issuing = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES; clock = kernel cycles / time.
valu_issue / salu_issue = SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_SCA x 4 cycles over (1024 SIMDs x kernel cycles) -- the share of a SIMD's time its vector
(incl. the MFMAs' issue slots) / scalar issue port is taken (SQ_ACTIVE_* and SQ_WAVE_CYCLES count in units of 4 cycles: ~1.03 per VALU instruction);
waves = SQ_WAVE_CYCLES x 4 / (1024 x kernel cycles), the average resident waves per SIMD."""
import collections, csv, re, sys


def load(path):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
        d[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return d, dur


a, dur = load(sys.argv[1])
b, _ = load(sys.argv[2])
print("\n# derived (batch 32; see tools/pmc_derive.py for the formulas)")
for k in sorted(dur, key=lambda k: -dur[k])[:40]:
    if "dffw::" not in k:
        continue
    ca, cb = a[k], b[k]
    cyc = ca.get("GRBM_GUI_ACTIVE", 0) / 8.0
    if cyc <= 0 or ca.get("SQ_WAVE_CYCLES", 0) <= 0:
        continue
    wc = ca["SQ_WAVE_CYCLES"]
    lds_idx = cb.get("SQ_LDS_IDX_ACTIVE", 0)
    print(f"#   {k:58s} time {dur[k]:7.0f} us  clock {cyc / dur[k] / 1e3:4.2f} GHz  mfma_pipe {ca['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):4.2f}  "
          f"lds_busy {lds_idx / (256 * cyc):4.2f}  lds_conflict {(cb.get('SQ_LDS_BANK_CONFLICT', 0) / lds_idx) if lds_idx else 0:4.2f}  "
          f"waiting {ca['SQ_WAIT_ANY'] / wc:4.2f}  stalled {ca['SQ_WAIT_INST_ANY'] / wc:4.2f}  valu_issue {cb.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (1024 * cyc):4.2f}  "
          f"salu_issue {cb.get('SQ_ACTIVE_INST_SCA', 0) * 4 / (1024 * cyc):4.2f}  waves {wc * 4 / (1024 * cyc):3.1f}")
