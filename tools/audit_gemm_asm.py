"""Build-time check of the ring-fed GEMM kernels (csrc/gemm_rs.hip, csrc/gemm_os.hip) on the compiler's assembly.

Both kernels run one wave per SIMD with most of the 512 registers spoken for (the token panel, or 256 accumulators).
Small source changes have made hipcc put the accumulators in vector registers and spill everything else, which costs
a factor of 4-5 and breaks the counted waits (spill reloads are vector-memory operations the kernels do not count).
Fails the build when a kernel of the file uses scratch memory.

    python tools/audit_gemm_asm.py build/gemm_os.s
"""
import re
import sys

txt = open(sys.argv[1]).read()
bad = []
n = 0
for m in re.finditer(r"\.set\s+(\S*gemm_(?:rs|os)\S*)\.private_seg_size,\s*(\d+)", txt):
    n += 1
    if int(m.group(2)) != 0:
        bad.append((m.group(1), int(m.group(2))))
if n == 0:
    sys.exit(f"{sys.argv[1]}: no gemm_rs / gemm_os kernel found")
if bad:
    sys.exit("scratch (register spills) in: " + ", ".join(f"{k} ({b} B/lane)" for k, b in bad))
print(f"{sys.argv[1]}: {n} kernels, no scratch")
