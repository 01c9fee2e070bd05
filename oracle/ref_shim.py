"""
TEST INFRASTRUCTURE ONLY -- imports the UNMODIFIED reference from /root/reference.

Works only in the build container (the reference tree does not travel to the
GPU box); used by ``oracle/make_golden.py`` to generate ``tests/golden/*`` and by
the optional ``reference``-marked tests.  The reference's ``self_play.py`` imports
``ray`` (not installable here) only for the ``@ray.remote`` decorator and actor
plumbing; a 3-line stub makes ``SelfPlay`` a plain class.
``gym`` / ``cv2`` are stubbed so game files whose ``MuZeroConfig`` we need can be
imported (their ``Game`` classes are not used).
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MZX_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "self_play.py"))


def load():
    """Return (models, self_play) modules of the reference."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    if "ray" not in sys.modules:
        ray = types.ModuleType("ray")
        ray.remote = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda c: c))
        ray.get = lambda x: x
        sys.modules["ray"] = ray
    for name in ("gym", "cv2"):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except ImportError:
                sys.modules[name] = types.ModuleType(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    models = importlib.import_module("models")
    self_play = importlib.import_module("self_play")
    return models, self_play


def game_module(name):
    load()
    return importlib.import_module("games." + name)


def muzero_config(name):
    """
    ``games/<name>.py``'s MuZeroConfig().  Game files whose environment needs a package that is not installed
    (lunarlander: Box2D) cannot be imported as modules; their UNMODIFIED ``MuZeroConfig`` class definition is
    then taken from the file's syntax tree and executed on its own (it depends only on datetime / pathlib / torch).
    """
    try:
        return game_module(name).MuZeroConfig()
    except ImportError:
        import ast
        path = os.path.join(REFERENCE_ROOT, "games", name + ".py")
        with open(path) as f:
            tree = ast.parse(f.read(), filename=path)
        keep = [n for n in tree.body
                if (isinstance(n, ast.ClassDef) and n.name == "MuZeroConfig")
                or (isinstance(n, (ast.Import, ast.ImportFrom)) and
                    all(a.name.split(".")[0] in ("datetime", "pathlib", "os", "math", "numpy", "torch") for a in n.names)
                    and getattr(n, "module", None) in (None, "datetime", "pathlib", "os", "math"))]
        ns = {"__file__": path, "__name__": "games." + name}
        exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
        return ns["MuZeroConfig"]()
