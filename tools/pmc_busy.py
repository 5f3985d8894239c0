"""Per-kernel busy figures from rocprofv3 --pmc result databases of tools/pmc_workload.py.
SQ_ACTIVE_INST_* count quad-cycles summed over the chip, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs:
  VALU busy = 4 SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
  LDS issue = 4 SQ_ACTIVE_INST_LDS  / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)   (LDS instructions in flight, per SIMD)
  LDS pipe  = SQ_LDS_IDX_ACTIVE     / (GRBM_GUI_ACTIVE / 8 * 256 CUs)      (the CU's LDS index/data pipeline active)
  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) [as reported, x4 if quad-cycles]
usage: pmc_busy.py <dir with one sub-directory per pass> [out.json]   (the JSON is what bench.py reads for roofline.binding)"""
import collections, glob, json, os, re, sqlite3, sys


def lib_identity():
    """po_version() of the library in this tree - the one the passes just profiled (tools/profile_round.sh runs this script on
    the same box, same snapshot) - and the source hash inside it (tools/source_hash.py)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from phyloligo_amd import _lib
    v = _lib.load().po_version().decode()
    return v, v.rsplit("src ", 1)[-1] if "src " in v else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)\(", name)
    return m.group(1) if m else name[:70]


vals = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)):
    c = sqlite3.connect(f)
    # the largest dispatch of each kernel (the N = 50 000 tile launch that did the work)
    for name, counter, val in c.execute("select kernel_name, counter_name, max(value) from counters_collection group by kernel_name, counter_name"):
        vals[short(name)][counter] = val
KEYS = {"jsd_lut_rows_kernel<double, 16>": "jsd_lut_rows_kernel", "valu_tile_kernel<1, double, 4>": "valu_tile_kernel<JSD>",
        "valu_tile_kernel<3, double, 8>": "valu_tile_kernel<BC>", "gram_i8_quad_kernel<0, double>": "gram_i8_quad_kernel<f64>",
        "gram_tile_kernel<0, double>": "gram_tile_kernel<f64>", "bc_sad_tile_kernel<double>": "bc_sad_tile_kernel",
        "pairdot_tile_kernel<1, 0, double>": "pairdot_tile_kernel<KT>", "pairdot_tile_kernel<1, 1, double>": "pairdot_tile_kernel<BC>",
        "gram_i8_half_kernel<4, double>": "gram_i8_half_kernel<SC,f64>", "gram_i8_half_kernel<0, double>": "gram_i8_half_kernel<f64>",
        "gram_i8_quad_kernel<0, float>": "gram_i8_quad_kernel<f32>", "gram_i8_half_kernel<0, float>": "gram_i8_half_kernel<f32>",
        "gram_i8_half_kernel<4, float>": "gram_i8_half_kernel<SC,f32>", "bc_sad_tile_kernel<float>": "bc_sad_tile_kernel<f32>"}
summary = {}
print("%-44s %10s %9s %9s %9s %9s %12s %10s" % ("kernel", "GUI cyc/8", "VALU %", "LDSissue%", "LDSpipe %", "MFMA %", "LDS confl %", "waves"))
for k, v in sorted(vals.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    gui = v.get("GRBM_GUI_ACTIVE", 0) / 8
    if gui < 1e5: continue
    f = lambda x: ("%9.1f" % x) if x is not None else "        -"
    valu = 400 * v["SQ_ACTIVE_INST_VALU"] / (gui * 1024) if "SQ_ACTIVE_INST_VALU" in v else None
    lds = 400 * v["SQ_ACTIVE_INST_LDS"] / (gui * 1024) if "SQ_ACTIVE_INST_LDS" in v else None
    pipe = 100 * v["SQ_LDS_IDX_ACTIVE"] / (gui * 256) if "SQ_LDS_IDX_ACTIVE" in v else None
    mfma = 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024) if "SQ_VALU_MFMA_BUSY_CYCLES" in v else None
    confl = 100 * v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else None
    print("%-44s %10.3e %s %s %s %s %s %10.0f" % (k[:44], gui, f(valu), f(lds), f(pipe), f(mfma), "   " + f(confl), v.get("SQ_WAVES", 0)))
    if k in KEYS:
        r = lambda x: None if x is None else round(x / 100.0, 4)
        summary[KEYS[k]] = {"valu": r(valu), "lds": r(pipe), "mfma": r(mfma), "lds_conflict": r(confl), "kernel": k}
    extra = {c: v[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_I8", "SQ_INSTS_VALU_MFMA_MOPS_F64") if v.get(c)}
    if extra: print("      " + "  ".join("%s=%.4g" % kv for kv in extra.items()))

if len(sys.argv) > 2:
    summary["_lib_version"], summary["_src_hash"] = lib_identity()
    summary["_source"] = "rocprofv3 --pmc passes over tools/pmc_workload.py (N = 50 000), summarised by tools/pmc_busy.py; busy = fraction of the kernel's GRBM_GUI_ACTIVE time"
    with open(sys.argv[2], "w") as fh:
        json.dump(summary, fh, indent=1)
