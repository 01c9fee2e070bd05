"""
TEST / BASELINE INFRASTRUCTURE ONLY -- recipe for ``oracle/_ref/``.

Compiles the UNMODIFIED reference modules of the hot path, from the sources where they lie under
/root/reference (``self_play.py``: MCTS / Node / MinMaxStats / GameHistory / SelfPlay, and ``models.py``), to
CPython bytecode files ``oracle/_ref/{self_play,models}.pyc`` -- compiled output only: no reference source is
copied into the repository, ``oracle/_ref/`` is git-ignored (it travels to the GPU box with the snapshot
like the built ``.so`` files, since /root/reference does not exist there).

Who may use it (oracle/ rule): ``bench.py``'s ``cpu_baseline`` leg, which times the reference's own
``MCTS(config).run`` on the GPU box's host cores (SURVEY.md section 8d asks for the unmodified reference as the
CPU baseline, ``kind: "reference"``), and tests.  The product never imports it.

Run by ``__graft_entry__.build()`` whenever /root/reference is present (the build container).
"""
import importlib.machinery
import importlib.util
import os
import py_compile
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
MODULES = ("models", "self_play")


def build(reference_root="/root/reference", force=False):
    """Returns the list of .pyc files written (empty when the reference tree is absent)."""
    if not os.path.isfile(os.path.join(reference_root, "self_play.py")):
        return []
    os.makedirs(OUT, exist_ok=True)
    written = []
    for name in MODULES:
        src, dst = os.path.join(reference_root, name + ".py"), os.path.join(OUT, name + ".pyc")
        if force or not os.path.isfile(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            py_compile.compile(src, cfile=dst, dfile=f"<reference>/{name}.py", doraise=True, optimize=0)
        written.append(dst)
    with open(os.path.join(OUT, "PYTHON_VERSION"), "w") as f:
        f.write("%d.%d\n" % sys.version_info[:2])   # bytecode is only valid for the interpreter that wrote it
    return written


def available():
    if not all(os.path.isfile(os.path.join(OUT, m + ".pyc")) for m in MODULES):
        return False
    try:
        with open(os.path.join(OUT, "PYTHON_VERSION")) as f:
            return f.read().strip() == "%d.%d" % sys.version_info[:2]
    except OSError:
        return False


def load():
    """(models, self_play) modules of the unmodified reference, from the compiled files (``ray`` stubbed)."""
    if not available():
        raise RuntimeError("oracle/_ref is not built (python oracle/build_ref.py in the build container)")
    if "ray" not in sys.modules:   # self_play.py only uses ray for the @ray.remote decorator / actor plumbing
        ray = types.ModuleType("ray")
        ray.remote = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda c: c))
        ray.get = lambda x: x
        sys.modules["ray"] = ray
    mods = []
    for name in MODULES:
        qual = "mzx_ref_" + name
        if qual not in sys.modules:
            loader = importlib.machinery.SourcelessFileLoader(qual, os.path.join(OUT, name + ".pyc"))
            spec = importlib.util.spec_from_loader(qual, loader)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[qual] = mod
            if name == "models":
                sys.modules.setdefault("models", mod)   # self_play.py does `import models`
            loader.exec_module(mod)
        mods.append(sys.modules[qual])
    return tuple(mods)


if __name__ == "__main__":
    print("\n".join(build(force="--force" in sys.argv)) or "reference tree not present: nothing built")
