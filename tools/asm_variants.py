"""Generate code objects of assembly-kernel variants for tools/asm_bench.py:
    python tools/asm_variants.py outdir name[:opt=val,...] ...     (name 'diag...' -> stamped build)"""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "grape.jl_amd", "csrc", "asm"))
import gen_t16
out = sys.argv[1]
os.makedirs(out, exist_ok=True)
llvm = "/opt/rocm/lib/llvm/bin"
for spec in sys.argv[2:]:
    name, _, o = spec.partition(":")
    opts = {}
    for kv in filter(None, o.split(",")):
        k, _, v = kv.partition("=")
        opts[k] = int(v) if v.lstrip("-").isdigit() else (v or True)
    stop = opts.pop("stop", None)
    g, prog, text = gen_t16.generate(os.path.join(out, name + ".s"), diag=name.startswith("diag"), stop_after=stop, opts=opts)
    subprocess.run([f"{llvm}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", os.path.join(out, name + ".s"), "-o", os.path.join(out, name + ".o")], check=True)
    subprocess.run([f"{llvm}/ld.lld", "-shared", os.path.join(out, name + ".o"), "-o", os.path.join(out, name + ".co")], check=True)
    os.remove(os.path.join(out, name + ".o"))
    body = prog.ins[prog.labels["L_cell"]:]
    print(f"{name}: {sum(1 for i in body if i.kind == 'mfma')} mfma, {sum(1 for i in body if i.kind in ('valu', 'dpp', 'rdlane'))} valu, "
          f"{sum(1 for i in body if i.kind == 'lds')} lds, {sum(i.src[0] + 1 for i in body if i.kind == 'nop')} nop states per cell")
