"""
Random OBJ / MTL text through the C++ host loader (host/scene_loader.cpp ≙ Scene::load, src/scene.rs:95-207 + tobj 4.0.3)
and through its Python restatement (oracle/obj_loader.py): the vertex / index / geometry / model arrays must hold the same bytes
and the object names must agree.  CPU only.  What is drawn: several objects (`o` / `g`), `usemtl` changes inside an object
(tobj starts a new model), polygons of 3-7 corners (fan triangulation), `v//n` and `v/t/n` corners, negative (relative) indices,
corner de-duplication, number spellings (`1`, `-0.50`, `+3.`, `.25`, `1e-3`, `2.5E+1`), comments, blank lines, tabs, trailing blanks,
CRLF line ends, MTL records with Kd / Ks / Ke / Pm / Pr in any order and unknown keys.
"""
import numpy as np
import pytest


def _num(rng):
    x = float(rng.choice([rng.uniform(-10, 10), rng.integers(-5, 6), rng.uniform(-1, 1) * 10.0 ** rng.integers(-6, 4)]))
    style = int(rng.integers(0, 6))
    if style == 0:
        return f"{x:.6g}"
    if style == 1:
        return f"{x:.3f}"
    if style == 2:
        return f"{x:.4e}".replace("e", str(rng.choice(["e", "E"])))
    if style == 3:
        return ("+" if x >= 0 else "") + f"{x:.2f}"
    if style == 4:
        s = f"{x:.3f}"
        return s.replace("0.", ".", 1) if abs(x) < 1 else s
    return str(int(x)) + str(rng.choice(["", ".", ".0"]))


def _draw(rng):
    sep = lambda: str(rng.choice([" ", "  ", "\t", " \t "]))
    eol = str(rng.choice(["\n", "\r\n"]))
    n_mat = int(rng.integers(1, 4))
    mtl = ["# materials" + eol]
    for m in range(n_mat):
        mtl.append(f"newmtl mat{m}" + eol)
        keys = [k for k in ["Kd", "Ks", "Ke", "Pm", "Pr", "Ns", "illum", "Xq"] if rng.random() < 0.7]
        rng.shuffle(keys)
        for k in keys:
            n = 3 if k in ("Kd", "Ks", "Ke") else 1
            mtl.append(k + sep() + sep().join(_num(rng) if k != "illum" else "2" for _ in range(n)) + str(rng.choice(["", " "])) + eol)
        if rng.random() < 0.3:
            mtl.append(eol)
    obj = ["mtllib s.mtl" + eol]
    n_pos = n_nrm = n_tex = 0
    for o in range(int(rng.integers(1, 5))):
        if rng.random() < 0.9 or o > 0:
            obj.append(str(rng.choice(["o", "g"])) + sep() + f"object {o}" + str(rng.choice(["", " "])) + eol)
        nv, nn, nt = int(rng.integers(3, 12)), int(rng.integers(1, 5)), int(rng.integers(1, 5))
        with_tex = rng.random() < 0.5
        lines = [f"v{sep()}{_num(rng)}{sep()}{_num(rng)}{sep()}{_num(rng)}" for _ in range(nv)] + \
                [f"vn{sep()}{_num(rng)}{sep()}{_num(rng)}{sep()}{_num(rng)}" for _ in range(nn)] + \
                ([f"vt{sep()}{_num(rng)}{sep()}{_num(rng)}" for _ in range(nt)] if with_tex else [])
        rng.shuffle(lines)
        for l in lines:
            obj.append(l + str(rng.choice(["", " ", "\t"])) + eol)
            if rng.random() < 0.1:
                obj.append(str(rng.choice(["", "# a comment", "   "])) + eol)
        p0, n0, t0 = n_pos, n_nrm, n_tex
        n_pos, n_nrm, n_tex = n_pos + nv, n_nrm + nn, n_tex + (nt if with_tex else 0)
        obj.append(f"usemtl mat{int(rng.integers(0, n_mat))}" + eol)
        for f in range(int(rng.integers(1, 8))):
            if rng.random() < 0.2:
                obj.append(f"usemtl mat{int(rng.integers(0, n_mat))}" + eol)      # may or may not change the material
            corners = []
            for _ in range(int(rng.integers(3, 8))):
                relative = rng.random() < 0.3

                def ref(lo, hi, total):
                    i = int(rng.integers(lo, hi))                                  # 0-based absolute
                    return str(i - total) if relative else str(i + 1)
                v, n = ref(p0 if rng.random() < 0.8 else 0, n_pos, n_pos), ref(n0 if rng.random() < 0.8 else 0, n_nrm, n_nrm)
                corners.append(f"{v}/{ref(t0, n_tex, n_tex)}/{n}" if with_tex else f"{v}//{n}")
            obj.append("f" + sep() + sep().join(corners) + eol)
    return "".join(obj), "".join(mtl)


@pytest.mark.parametrize("seed", range(8))
def test_random_obj_text_loads_the_same_bytes(rt, obj_loader, tmp_path, seed):
    rng = np.random.default_rng(1000 + seed)
    for case in range(60):
        obj, mtl = _draw(rng)
        d = tmp_path / f"c{case}"
        d.mkdir()
        (d / "s.mtl").write_bytes(mtl.encode())
        (d / "s.obj").write_bytes(obj.encode())
        path = str(d / "s.obj")
        want = obj_loader.load(path)
        got = rt.HostScene(path)
        for k in ["vertices", "indices", "primitives", "models"]:
            a, b = getattr(got, k), getattr(want, k)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8)), f"seed {seed} case {case}: {k} differ\n{obj}\n{mtl}"
        assert got.names == want.names, f"seed {seed} case {case}: names {got.names} vs {want.names}\n{obj}"


@pytest.mark.parametrize("seed", range(4))
def test_broken_obj_text_is_refused_by_both(rt, obj_loader, tmp_path, seed):
    """One defect per file — no material for an object, a corner without a normal, an index beyond the arrays, a face of two corners,
    an unknown material name: where the reference would panic (src/scene.rs:121-124,149; tobj) both loaders refuse the file."""
    rng = np.random.default_rng(2000 + seed)
    checked = 0
    for case in range(40):
        obj, mtl = _draw(rng)
        lines = obj.splitlines(keepends=True)
        kind = int(rng.integers(0, 5))
        faces = [i for i, l in enumerate(lines) if l.startswith("f")]
        if kind == 0:
            lines = [l for l in lines if not l.startswith("usemtl")]
        elif kind == 1:
            i = faces[int(rng.integers(0, len(faces)))]
            toks = lines[i].split()
            toks[1] = toks[1].split("/")[0]                       # "v" alone: no normal
            lines[i] = " ".join(toks) + "\n"
        elif kind == 2:
            i = faces[int(rng.integers(0, len(faces)))]
            toks = lines[i].split()
            parts = toks[2].split("/")
            parts[0] = "99999"
            toks[2] = "/".join(parts)
            lines[i] = " ".join(toks) + "\n"
        elif kind == 3:
            i = faces[int(rng.integers(0, len(faces)))]
            lines[i] = " ".join(lines[i].split()[:3]) + "\n"      # two corners
        else:
            lines = [("usemtl nosuchmaterial\n" if l.startswith("usemtl") else l) for l in lines]
        d = tmp_path / f"b{case}"
        d.mkdir()
        (d / "s.mtl").write_bytes(mtl.encode())
        (d / "s.obj").write_bytes("".join(lines).encode())
        path = str(d / "s.obj")
        try:
            obj_loader.load(path)
            python_ok = True
        except (ValueError, IndexError, KeyError):
            python_ok = False
        try:
            rt.HostScene(path)
            cpp_ok = True
        except rt.RtError:
            cpp_ok = False
        assert python_ok == cpp_ok, f"seed {seed} case {case} defect {kind}: python {'loads' if python_ok else 'refuses'}, C++ {'loads' if cpp_ok else 'refuses'}\n{''.join(lines)}"
        checked += 0 if python_ok else 1
    assert checked >= 30
