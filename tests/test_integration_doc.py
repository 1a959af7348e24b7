"""CPU: the Rust `extern "C"` block in INTEGRATION.md agrees with include/atlas_hip.h (names, parameter counts,
usize vs c_int returns)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rust_declarations_match_the_header():
    h = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "atlas_hip.h")).read(), flags=re.S)
    c = {}
    for ret, name, params in re.findall(r"\b([a-z_ ]*?[a-z_0-9]+\s*\**)\s*(atlas_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        params = " ".join(params.split())
        c[name] = (0 if params in ("", "void") else len(params.split(",")), ret.strip())
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    decls = re.findall(r"pub fn (atlas_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", doc, flags=re.S)
    assert len(decls) >= 60
    for name, params, ret in decls:
        assert name in c, name
        n = len([p for p in " ".join(params.split()).split(",") if p.strip()])
        assert n == c[name][0], (name, n, c[name][0])
        assert ("size_t" in c[name][1]) == ((ret or "").strip() == "usize"), (name, ret, c[name][1])
