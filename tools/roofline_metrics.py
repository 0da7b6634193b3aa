"""Derived per-kernel figures from the raw rocprofv3 counters collected by tools/profile_round.sh (one counter group per
run).  Usage: python tools/roofline_metrics.py gpurun_out/<tag>_roofline.json  -> rewrites the file with `derived` blocks.

Normalisation on MI355X (8 XCDs x 4 SEs, 256 CUs x 4 SIMDs; checked on k_vm_bwd_brick: 16.15 M v_mfma_f32_16x16x4_f32 x 32
cycles/SIMD = 516.8 M = SQ_VALU_MFMA_BUSY_CYCLES exactly):
  GRBM_GUI_ACTIVE            summed over the 8 XCDs          -> active cycles of the launch = GRBM_GUI_ACTIVE / 8
  SQ_VALU_MFMA_BUSY_CYCLES   cycles, summed over 1024 SIMDs  -> mfma_busy = it / (1024 * active cycles)
  SQ_ACTIVE_INST_VALU        quad-cycles (4 clocks) of VALU ISSUE, summed over SIMDs; an MFMA counts as one quad-cycle
                             here whatever its length (k_brdf_mlp_bwd: 16.7 M for 16.6 M VALU instructions of which
                             2.7 M are 64-cycle MFMAs)       -> valu_issue_busy = 4 * it / (1024 * active cycles)
  alu_busy                   = mfma_busy + valu_issue_busy - 4 * SQ_INSTS_MFMA / (1024 * active cycles).  The two ADD UP on
                             gfx950: while an fp32 MFMA executes, no other VALU instruction of any wave issues on that SIMD
                             (tools/ub/coexec.hip: an MFMA wave and an FMA wave on one SIMD take longer than one after the
                             other; SQ_VALU_MFMA_COEXEC_CYCLES = 0 in every kernel here) -- the fp32 matrix rate equals the
                             packed-FMA rate, so the ceiling of an fp32-MFMA kernel is alu_busy = 1, not mfma_busy = 1.
  SQ_WAVE_CYCLES             quad-cycles of resident waves   -> occupancy = 4 * it / (1024 * active cycles) waves per SIMD
  TCC_REQ_sum                128-byte L2 requests            -> l2_frac against 34.5 TB/s
  FETCH_SIZE / WRITE_SIZE    KB at the fabric (uncorrected, see MI355X_MICROARCH.md "HBM") -> hbm_frac against 8 TB/s
"""
import json
import sys

HBM, L2, SIMDS = 8.0e12, 34.5e12, 1024


def derive(name, rec):
    c = rec.get("counters_per_launch", {})
    t = rec.get("avg_launch_us")
    if not c or not t:
        return rec
    t *= 1e-6
    d = {}
    act = c.get("GRBM_GUI_ACTIVE")
    cyc = act / 8 if act else None
    if cyc:
        d["active_cycles"] = round(cyc)
        d["clock_GHz"] = round(cyc / t / 1e9, 2)
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            d["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc), 4)
        if c.get("SQ_ACTIVE_INST_VALU") is not None:
            d["valu_issue_busy"] = round(4 * c["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc), 4)
            d["alu_busy"] = round(d.get("mfma_busy", 0.0) + d["valu_issue_busy"] - 4 * c.get("SQ_INSTS_MFMA", 0) / (SIMDS * cyc), 4)
        if c.get("SQ_WAVE_CYCLES") is not None:
            d["waves_per_simd"] = round(4 * c["SQ_WAVE_CYCLES"] / (SIMDS * cyc), 2)
        if c.get("SQ_WAIT_INST_ANY") is not None and c.get("SQ_WAVE_CYCLES"):
            d["issue_stall_frac_of_wave_cycles"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        if c.get("SQ_WAIT_ANY") is not None and c.get("SQ_WAVE_CYCLES"):
            d["waitcnt_frac_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    if c.get("FETCH_SIZE") is not None and c.get("WRITE_SIZE") is not None:
        b = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        d["hbm_bytes"] = round(b)
        d["hbm_frac"] = round(b / t / HBM, 4)
    if c.get("TCC_REQ_sum") is not None:
        d["l2_frac"] = round(c["TCC_REQ_sum"] * 128 / t / L2, 4)
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None:
            d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), 4)
    if c.get("SQ_INSTS_MFMA"):
        flop = c["SQ_INSTS_MFMA"] * (4096 if "brdf_mlp" in name else 2048)     # 32x32x2 / 16x16x4 f32-input MFMA
        d["mfma_tflops"] = round(flop / t / 1e12, 2)
        d["mfma_frac_of_157.3_tflops"] = round(flop / t / 157.3e12, 4)
    fr = {k: d[k] for k in ("alu_busy", "hbm_frac", "l2_frac") if k in d}
    if fr:
        d["bound"] = max(fr, key=fr.get)
        d["frac"] = fr[d["bound"]]
    out = {k: v for k, v in rec.items() if k in ("avg_launch_us", "launches_profiled", "counters_per_launch", "note",
                                                 "hbm_bytes_per_launch")}
    out["derived"] = d
    if "hbm_bytes" in d:
        out["hbm_bytes_per_launch"] = d["hbm_bytes"]
    return out


def main(path):
    doc = json.load(open(path))
    doc["kernels"] = {k: derive(k, v) for k, v in doc["kernels"].items()}
    doc["normalisation"] = __doc__.split("Normalisation", 1)[1].strip()
    json.dump(doc, open(path, "w"), indent=1)
    for k, v in doc["kernels"].items():
        if "derived" in v:
            print(f"{k:30s} {v['avg_launch_us']:8.1f} us  " + "  ".join(f"{a}={b}" for a, b in v["derived"].items()
                                                                            if a not in ("active_cycles", "hbm_bytes")))


if __name__ == "__main__":
    main(sys.argv[1])
