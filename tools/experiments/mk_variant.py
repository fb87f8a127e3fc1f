#!/usr/bin/env python3
"""Build a VARIANT of libcdml_hip.so from a scratch copy of csrc/ with literal text replacements applied --
kernel A/B runs and timing ablations without touching the product sources.

usage: python tools/experiments/mk_variant.py TAG RECIPE.py [RECIPE2.py ...]
  RECIPE.py defines EDITS = [(file, old, new), ...] (old must occur exactly once unless (file, old, new, count));
  the result is build/variants/libcdml_TAG.so; run with CDML_LIB_PATH=build/variants/libcdml_TAG.so.
"""
import os
import runpy
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

tag, recipes = sys.argv[1], sys.argv[2:]
scratch = os.path.join(ROOT, "build", "variants", "src_" + tag)
shutil.rmtree(scratch, ignore_errors=True)
shutil.copytree(ge.CSRC, scratch)
for r in recipes:
    for e in runpy.run_path(r)["EDITS"]:
        f, old, new = e[:3]
        want = e[3] if len(e) > 3 else 1
        p = os.path.join(scratch, f)
        s = open(p).read()
        n = s.count(old)
        if n != want:
            sys.exit("%s: %r occurs %d times in %s (expected %d)" % (r, old[:60], n, f, want))
        open(p, "w").write(s.replace(old, new))
lib = os.path.join(ROOT, "build", "variants", "libcdml_%s.so" % tag)
objdir = os.path.join(ROOT, "build", "variants", "obj_" + tag)
shutil.rmtree(objdir, ignore_errors=True)
base = os.path.join(ROOT, "build", "obj")               # objects of the unedited sources: reused (same mtimes)
if os.path.isdir(base):
    shutil.copytree(base, objdir)
if os.path.exists(lib):
    os.remove(lib)
ge.build(csrc=scratch, lib=lib, objdir=objdir)
print("built", lib)
