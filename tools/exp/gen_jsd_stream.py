"""Generator of experimental instruction streams for the word pair of jsd_lut_rows_kernel (the shipped stream is written with
macros in csrc/po_jsd_lut.hip).  usage: gen_jsd_stream.py <depth 2|3> <interleave 0|1> <consume_first 0|1> > stream.inc"""
import sys
depth, inter, cfirst = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
def issue(W, T, ra, rb, B0, B1):
    pairs = [(ra, B0), (ra, B1), (rb, B0), (rb, B1)]
    if inter:
        out = []
        for i, (r, b) in enumerate(pairs):
            out += ['v_add_u32 %%[t0], %%[%s%d], %%[%s]' % (W, r, b), 'ds_read_b64 %%[%s%d], %%[t0]' % (T, i)]
        return out
    return ['v_add_u32 %%[t%d], %%[%s%d], %%[%s]' % (i, W, r, b) for i, (r, b) in enumerate(pairs)] + \
           ['ds_read_b64 %%[%s%d], %%[t%d]' % (T, i, i) for i in range(4)]
def add(T, base):
    return ['v_add_f64 %%[c%d], %%[c%d], %%[%s%d]' % (base + i, base + i, T, i) for i in range(4)]
bufs = "nmp"[:depth]
lines = []
for g in range(16):
    W, B0, B1 = ("a", "b0", "b1") if g < 8 else ("e", "d0", "d1")
    r = 2 * (g % 8)
    iss = issue(W, bufs[g % depth], r, r + 1, B0, B1)
    h = g - (depth - 1)
    if h < 0:
        lines += iss
    elif cfirst and depth == 2:
        # consume group h before issuing group g: everything issued so far may have to be back but the last (depth-2)*4
        lines += ['s_waitcnt lgkmcnt(0)'] + add(bufs[h % depth], 4 * (h % 8)) + iss
    else:
        lines += iss + ['s_waitcnt lgkmcnt(%d)' % (4 * (depth - 1))] + add(bufs[h % depth], 4 * (h % 8))
for h in range(16 - (depth - 1), 16):
    lines += ['s_waitcnt lgkmcnt(%d)' % (4 * (15 - h))] + add(bufs[h % depth], 4 * (h % 8))
sys.stdout.write("".join('    "%s\\n\\t"\n' % l.replace('%%', '%') for l in lines))
