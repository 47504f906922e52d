#!/usr/bin/env python3
"""SHA-256 of the machine code of every kernel of csrc/libmpcgpu.so (branch targets included: position-independent within a kernel).
Why: the register allocator of this toolchain places saves of vector registers under divergent exec masks in a way that has corrupted lanes
in some builds and not in others ("Round 6"); a build is trusted when the GPU suite, tools/pipe_stress.py and
tools/help_check.py pass on it.  A later change that leaves a kernel's hash untouched leaves that validation standing for the kernel.
Usage: python tools/kernel_hashes.py [libmpcgpu.so] [--diff other_hashes.txt]      (build container: no GPU needed)"""
import hashlib, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(lib):
    d = tempfile.mkdtemp()
    tmp = os.path.join(d, "lib.so")
    os.symlink(os.path.abspath(lib), tmp)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", tmp], check=True, stdout=subprocess.DEVNULL, cwd=d)
    for f in os.listdir(d):
        if "gfx950" in f:
            return os.path.join(d, f)
    raise SystemExit("no gfx950 code object in " + lib)


def hashes(lib):
    co = code_object(lib)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co], check=True, capture_output=True, text=True).stdout
    out, cur, h, pcrel = {}, None, None, 0
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]* ?<(\S+)>:$", line.strip())
        if m:
            if cur is not None:
                out[cur] = h.hexdigest()[:16]
            cur, h = m.group(1), hashlib.sha256()
            continue
        if cur is not None and line.strip():
            # (pc-relative addresses of other symbols -- a callee, a table in .rodata --: `s_getpc_b64` followed by `s_add_u32 / s_addc_u32 ..., literal`;
            #  the literal changes when ANOTHER kernel changes its size, the code does not)
            if "s_getpc_b64" in line:
                pcrel = 3
            elif pcrel > 0:
                pcrel -= 1
                if re.search(r"\bs_addc?_u32\b.*0x[0-9a-fA-F]+", line):
                    line = re.sub(r"0x[0-9a-fA-F]+\s*//.*$", "PCREL", line)
            # (symbolic branch-target annotations and the instruction's ADDRESS in the trailing comment dropped -- a kernel that merely moved because another
            #  one changed its size keeps its hash; the encoding words stay, with them every offset and literal)
            h.update(re.sub(r"\s+", " ", re.sub(r"// [0-9A-Fa-f]+:", "//", re.sub(r"<[^>]*>", "", line.strip()))).encode())
    if cur is not None:
        out[cur] = h.hexdigest()[:16]
    names = subprocess.run(["c++filt"], input="\n".join(out), capture_output=True, text=True).stdout.split("\n")
    return {re.sub(r"\(anonymous namespace\)::|\(.*$", "", n).replace("void ", "").replace(" ", ""): v for n, v in zip(names, out.values())}


if __name__ == "__main__":
    argv = sys.argv[1:]
    ref_path = argv[argv.index("--diff") + 1] if "--diff" in argv else None
    args = [a for i, a in enumerate(argv) if not a.startswith("--") and not (i and argv[i - 1] == "--diff")]
    lib = args[0] if args else os.path.join(ROOT, "motion-planning-for-autonomous-driving-with-mpc_amd", "csrc", "libmpcgpu.so")
    hs = hashes(lib)
    if "--diff" in sys.argv:
        ref = dict(l.split() for l in open(ref_path) if l.strip() and not l.startswith("#"))
        changed = [k for k in hs if ref.get(k) != hs[k]]
        print("kernels whose machine code differs from the reference list:", ", ".join(sorted(changed)) if changed else "none")
    else:
        for k in sorted(hs):
            print(k, hs[k])
