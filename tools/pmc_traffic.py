"""HBM-side traffic of every stage-2 tile kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes) over
tools/pmc_workload.py (one launch of every kernel at N = 50 000).  Per kernel the largest dispatch is taken (the tile launch
that did the work).  FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide streaming reads at 64 bytes, so it is doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact for 16-byte
streaming stores.
usage: pmc_traffic.py <dir with the pass databases> <source label> [traffic.json]   (the JSON is what bench.py reads for roofline.traffic)"""
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
    for name, counter, val in c.execute("select kernel_name, counter_name, max(value) from counters_collection group by kernel_name, counter_name"):
        vals[short(name)][counter] = val
N, PAIRS = 50000, 50000 * 49999 / 2.0
# kernel -> (key of traffic.json = "<metric>_n<N>_d<dim>" as bench.py looks it up, word-space size used for the algorithmic bytes)
KEYS = {"jsd_lut_rows_kernel<double, 16>": ("JSD_n50000_d256", 256), "gram_i8_quad_kernel<0, double>": ("Eucl_n50000_d256", 256),
        "gram_tile_kernel<0, double>": ("Eucl_f64_n50000_d256", 256), "pairdot_tile_kernel<1, 0, double>": ("KT_n50000_d256", 256),
        "pairdot_tile_kernel<1, 1, double>": ("BC_n50000_d4096", 4096), "valu_tile_kernel<1, double, 4>": ("JSD_ragged_n50000_d256", 256),
        "valu_tile_kernel<3, double, 8>": ("BC_ragged_n50000_d256", 256), "gram_i8_half_kernel<0, double>": ("Eucl_ragged_n50000_d256", 256),
        "bc_sad_tile_kernel<double>": ("BC_sad_n50000", 4096), "gram_i8_half_kernel<4, double>": ("SC_n50000_d256", 256),
        # float32 matrices (round 5): 8 bytes of output per pair
        "gram_i8_quad_kernel<0, float>": ("Eucl_f32_n50000_d256", 256, 8.0), "gram_i8_half_kernel<0, float>": ("Eucl_f32_ragged_n50000_d256", 256, 8.0),
        "gram_i8_half_kernel<4, float>": ("SC_f32_n50000_d256", 256, 8.0), "bc_sad_tile_kernel<float>": ("BC_sad_f32_n50000_d256", 256, 8.0)}
_ver, _hash = lib_identity()
out = {"_detail": {"how": __doc__.split("usage:")[0].strip(), "source": sys.argv[2], "lib_version": _ver, "src_hash": _hash}}
print("%-46s %14s %16s %14s %14s %8s" % ("kernel (largest dispatch)", "FETCH_SIZE KiB", "fetch B (x2)", "WRITE_SIZE KiB", "traffic B", "/ algo"))
for k, v in sorted(vals.items(), key=lambda kv: -(kv[1].get("WRITE_SIZE", 0) + kv[1].get("FETCH_SIZE", 0))):
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v or v["WRITE_SIZE"] + v["FETCH_SIZE"] < 1e5:
        continue
    fetch, write = 2.0 * v["FETCH_SIZE"] * 1024.0, v["WRITE_SIZE"] * 1024.0
    ratio = ""
    if k in KEYS:
        key, dim = KEYS[k][:2]
        algo = ((KEYS[k][2] if len(KEYS[k]) > 2 else 16.0) + 2.0 * dim * 4.0 / (N - 1)) * PAIRS
        ratio = "%.3f" % ((fetch + write) / algo)
        out[key] = fetch + write
        if key == "JSD_ragged_n50000_d256":      # the same kernel owns every tile of bench.py's uniform assembly under table_path=False
            out["JSD_general_n50000_d256"] = fetch + write
        out["_detail"][key] = {"kernel": k, "fetch_bytes_corrected": fetch, "write_bytes": write, "algorithmic_bytes": algo}
    print("%-46s %14.0f %16.4g %14.0f %14.4g %8s" % (k[:46], v["FETCH_SIZE"], fetch, v["WRITE_SIZE"], fetch + write, ratio))
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as fh:
        json.dump(out, fh, indent=1)
