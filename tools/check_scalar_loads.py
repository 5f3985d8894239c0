#!/usr/bin/env python3
"""Build-time check for jsd_lut_rows_kernel (csrc/po_jsd_lut.hip): its row counts are fetched by
`s_load_dwordx16` in one asm statement and waited for (`s_waitcnt lgkmcnt(0)`) in a LATER one, so between the two the
compiler believes the destination SGPRs already hold data.  Correctness then depends on register allocation: this script
compiles the file to gfx950 assembly with the flags of the Makefile and asserts that, in every instance of the kernel,
no instruction between such a load and the first full `lgkmcnt(0)` wait behind it reads or writes any of the 16
destination registers, and that no label or branch sits in between (straight-line code only, so the textual order is the
execution order).  Run by tests/test_host_cpu.py and by `make check-asm`.

usage: check_scalar_loads.py [po_jsd_lut.hip]      exit status 0 = ok
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "--offload-arch=gfx950", "-ffp-contract=off"]


def sregs(text):
    """SGPR numbers an operand list mentions: s5, s[4:7]"""
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def check(asm_text):
    problems, checked = [], 0
    for fm in re.finditer(r"^(_Z\w*jsd_lut_rows_kernel\w*):\s*(?:;.*)?$", asm_text, re.M):
        end = asm_text.index(".Lfunc_end", fm.end())
        lines = asm_text[fm.end():end].split("\n")
        for i, line in enumerate(lines):
            m = re.match(r"\s*s_load_dwordx16\s+s\[(\d+):(\d+)\],\s*s\[\d+:\d+\],\s*0x0\b", line)
            if not m:
                continue
            dest = set(range(int(m.group(1)), int(m.group(2)) + 1))
            checked += 1
            for j in range(i + 1, len(lines)):
                ins = lines[j].split(";")[0].strip()
                if not ins or ins.startswith("."):
                    if re.match(r"\.LBB\d+_\d+:", ins):
                        problems.append("%s: label %s between the load at +%d and its wait" % (fm.group(1)[:60], ins, i))
                        break
                    continue
                op = ins.split()[0]
                if op == "s_waitcnt" and re.search(r"lgkmcnt\(0\)", ins):
                    break
                if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                    problems.append("%s: %s between the load at +%d and its wait" % (fm.group(1)[:60], op, i))
                    break
                operands = ins[len(op):]
                if op.startswith("s_load_dword"):
                    # another load: its ADDRESS may reuse anything but our destination; its destination must be disjoint
                    pass
                if sregs(operands) & dest:
                    problems.append("%s: `%s` touches s[%d:%d] before the wait (load at +%d)"
                                    % (fm.group(1)[:60], ins, min(dest), max(dest), i))
                    break
            else:
                problems.append("%s: no lgkmcnt(0) wait behind the load at +%d" % (fm.group(1)[:60], i))
    return checked, problems


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "phyloligo_amd", "csrc", "po_jsd_lut.hip")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC] + FLAGS + ["-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL,
                       cwd=os.path.dirname(src))
        with open(out) as fh:
            text = fh.read()
    checked, problems = check(text)
    print("checked %d scalar row loads in jsd_lut_rows_kernel: %s" % (checked, "ok" if not problems else "%d PROBLEMS" % len(problems)))
    for p in problems:
        print("  " + p)
    return 1 if problems or checked == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
