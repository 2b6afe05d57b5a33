"""Build a VARIANT of the HIP library for an A/B on one GPU box (scripts/ab_libs.py, MVS_VIEWSEL_LIB): one source file with a textual
substitution, linked with the product's other objects.  usage: python scripts/build_variant.py NAME FILE 'old text' 'new text' [FILE old new ...]
-> mvs-texturing_amd/csrc/variants/libmvs_viewsel_NAME.so  (git-ignored, travels to the GPU box)
--patch=FILE.patch:TARGET.hip applies a unified diff to the copy of TARGET first (the sweep kernel's access-pattern probes live in
scripts/probe/sweep4_probes.patch, not in the product source):
    python scripts/build_variant.py cachehot --patch=scripts/probe/sweep4_probes.patch:k_mrf.hip -DMVS_SWEEP_EXP=1 k_mrf.hip '' ''"""
import os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mvs-texturing_amd"))
import build as B
name, rest = sys.argv[1], sys.argv[2:]
patches = [a[len("--patch="):].rsplit(":", 1) for a in rest if a.startswith("--patch=")]
rest = [a for a in rest if not a.startswith("--patch=")]
defines = [a for a in rest if a.startswith("-D")]            # e.g. -DMVS_SWEEP_EXP=1: rebuilds the files named FILE:-  (FILE '' '' keeps the text)
rest = [a for a in rest if not a.startswith("-D")]
assert len(rest) % 3 == 0 and rest
B.build_hip()
out_dir = os.path.join(B.CSRC, "variants"); os.makedirs(out_dir, exist_ok=True)
tmp = tempfile.mkdtemp(prefix="variant_", dir=B.CSRC)   # next to ctx.h / dmath.h so that relative includes resolve
try:
    objs = {s: os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.HIP_SOURCES}
    edits = {}
    for k in range(0, len(rest), 3):
        f, old, new = rest[k:k + 3]
        src = edits.get(f) or open(os.path.join(B.CSRC, f)).read()
        assert old in src, "text not found in %s: %r" % (f, old)
        edits[f] = src.replace(old, new) if old else src
    for f, src in edits.items():
        if f.endswith(".h"):
            raise SystemExit("header variants: edit every includer instead")
        p = os.path.join(B.CSRC, "_variant_" + name + "_" + f)
        open(p, "w").write(src)
        for pf, target in patches:
            if target == f:
                subprocess.check_call(["patch", "-s", p, os.path.join(ROOT, pf) if not os.path.isabs(pf) else pf])
        o = os.path.join(tmp, f.replace(".hip", ".o"))
        subprocess.check_call([B._hipcc()] + B.HIP_FLAGS + B.EXTRA_FLAGS.get(f, []) + defines + ["-c", "-x", "hip", p, "-o", o])
        os.remove(p)
        objs[f] = o
    lib = os.path.join(out_dir, "libmvs_viewsel_%s.so" % name)
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [objs[s] for s in B.HIP_SOURCES])
    print(lib)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
