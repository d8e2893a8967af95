"""Build check for the attention kernels that address accumulator registers by number (csrc/attn_impl.h): in the
compiler's assembly output, every v_accvgpr_* instruction OUTSIDE an inline-asm statement must name a register below
the range the kernel keeps for itself (forward: a64 and up; the backward kernels with 64 rows per wave: all of them), and the kernels
must use no scratch.

    python tools/audit_attn_asm.py file.s
"""
import re
import sys

RESERVED_FROM = {"attn_fwd_kernel": 64, "attn_bwd_dkv64_kernel": 0, "attn_bwd_dq64_kernel": 0}


def main(path):
    kernel, in_asm, bad, checked, first = None, False, [], 0, 0
    for n, line in enumerate(open(path), 1):
        m = re.match(r"^(_ZN\w*?(attn_fwd_kernel|attn_bwd_dkv64_kernel|attn_bwd_dq64_kernel)\w*):", line)
        if m:
            kernel, in_asm, first = m.group(1), False, RESERVED_FROM[m.group(2)]
            checked += 1
            continue
        if kernel is None:
            continue
        if ".end_amdhsa_kernel" in line or line.startswith(".Lfunc_end"):
            kernel = None
            continue
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        elif not in_asm:
            if "scratch_" in line:
                bad.append((n, line.strip()))
            if "v_accvgpr" in line:
                for a in re.findall(r"\ba\[?(\d+)", line):
                    if int(a) >= first:
                        bad.append((n, line.strip()))
    if checked == 0:
        print("audit_attn_asm: no audited kernel found in", path)
        return 1
    for n, l in bad[:20]:
        print(f"audit_attn_asm: {path}:{n}: {l}")
    print(f"audit_attn_asm: {checked} kernels, {len(bad)} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
