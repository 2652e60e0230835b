"""How the golden recipes import the REFERENCE -- test infrastructure (see oracle/__init__.py).

The repository ships an in-tree `models/hovernet/*` shim package (the import-by-name drop-in boundary) whose dotted
names equal the reference's.  A recipe that put the repository root ahead of /root/reference on sys.path would therefore import the
PRODUCT instead of the reference and regenerate "goldens" from the thing under test (round-2 verdict, weak #1).  Every recipe goes
through this module instead:

    use_reference(first=[...])   /root/reference at the head of sys.path (optional directories, e.g. a cv2 stand-in, ahead of it),
                                 the repository root at the TAIL: `hover_net_amd` (which exists only there) stays importable,
                                 `models.*`, `dataloader.*`, `misc.*`, `metrics.*`, `infer.*`, `run_utils.*` resolve to the reference
    ref_import(name)             import + assert that the module's file lies under /root/reference
    out_dir()                    tests/golden, or $HVN_GOLDEN_OUT (tests/test_golden_recipes.py regenerates into a temp dir)
    selected(names)              the cases to generate: all, or the comma list in $HVN_GOLDEN_CASES
"""
import importlib
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def use_reference(first=()):
    if not os.path.isdir(REF):
        raise RuntimeError("%s is missing: the golden recipes run in the build container only" % REF)
    drop = {os.path.realpath(p) for p in (REPO, REF, *first)}
    sys.path[:] = [p for p in sys.path if os.path.realpath(p or os.getcwd()) not in drop]
    sys.path[:0] = [*first, REF]
    sys.path.append(REPO)
    for name in list(sys.modules):          # a `models` (etc.) package imported earlier from the repository must not shadow the reference
        top = name.split(".")[0]
        f = getattr(sys.modules[name], "__file__", None) or ""
        if top in ("models", "dataloader", "misc", "metrics", "infer", "run_utils") and not os.path.realpath(f).startswith(REF + os.sep):
            del sys.modules[name]


def ref_import(name):
    m = importlib.import_module(name)
    f = os.path.realpath(getattr(m, "__file__", "") or "")
    if not f.startswith(REF + os.sep):
        raise ImportError("%s resolved to %s, not to the reference under %s" % (name, f, REF))
    return m


def out_dir():
    d = os.environ.get("HVN_GOLDEN_OUT") or os.path.join(REPO, "tests", "golden")
    os.makedirs(d, exist_ok=True)
    return d


def selected(names):
    want = os.environ.get("HVN_GOLDEN_CASES")
    if not want:
        return list(names)
    want = [w for w in want.split(",") if w]
    unknown = [w for w in want if w not in names]
    if unknown:
        raise KeyError("HVN_GOLDEN_CASES: unknown case(s) %s (known: %s)" % (unknown, list(names)))
    return want
