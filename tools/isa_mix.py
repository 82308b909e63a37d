"""Instruction mix of a kernel in a hipcc -S listing: python tools/isa_mix.py file.s [kernel-substring].
Counts by class (mfma / valu / LDS / waits / barriers / salu / VMEM) and the ratio to the MFMA count - the
issue-port budget per MFMA (MI355X_MICROARCH.md, row 'vector-instruction ISSUE cost') is what this is read against."""
import collections
import sys


def mix(path, sub="fused_blocks_kernel"):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sub in l.split(":")[0])
    end = next(i for i, l in enumerate(lines) if "s_endpgm" in l and i > start)
    c, vops = collections.Counter(), collections.Counter()
    for l in lines[start + 1:end]:
        l = l.strip()
        if not l or l.startswith((".", ";")) or l.split(";")[0].strip().endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"):
            k = "mfma"
        elif op.startswith("v_"):
            k = "valu"
            vops[op] += 1
        elif op.startswith("ds_") or op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"):
            k = op
        elif op.startswith("s_"):
            k = "salu"
        else:
            k = op
        c[k] += 1
    return c, vops


if __name__ == "__main__":
    c, vops = mix(sys.argv[1], *(sys.argv[2:3]))
    n = max(c["mfma"], 1)
    for k, v in c.most_common(40):
        print("%-28s %7d  %.2f per MFMA" % (k, v, v / n))
    print("VALU by opcode:", vops.most_common(30))
