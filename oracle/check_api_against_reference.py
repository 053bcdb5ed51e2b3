#!/usr/bin/env python3
"""Authoring container only (imports the reference by path): every PUBLIC name of the reference's hot-path modules -- planner.astar,
planner.differentiable_astar, planner.encoder, utils.data, utils.training -- must exist in this package's module of the same name, classes with
the same constructor parameters (names, order, defaults; this package may append more) and every public method likewise; constructed planners /
encoders carry every public instance attribute of the reference's objects and the same state_dict keys.  Dependencies the
container lacks (segmentation_models_pytorch, torchvision, pytorch_lightning, PIL, moviepy, pqdict) are stubbed: only signatures are read.
Prints one line per difference and a summary; exit code 1 on any difference that is not on the EXPECTED list (empty: pq_astar, out of scope, is a stub that raises)."""
import importlib
import importlib.util
import inspect
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.join(ROOT, "oracle")]
sys.path.insert(0, os.path.join(ROOT, "neural-astar_amd"))
REF = "/root/reference/src/neural_astar"
EXPECTED: set = set()  # (pq_astar exists as a stub that raises: out of scope)


class _Stub(types.ModuleType):
    __path__: list = []

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {"__init__": lambda self, *a, **kw: None, "__call__": lambda self, *a, **kw: self, "__getattr__": lambda self, n: self})


def main():
    for name in ("segmentation_models_pytorch", "torchvision", "torchvision.utils", "torchvision.transforms", "pytorch_lightning", "PIL", "PIL.Image",
                 "moviepy", "moviepy.editor", "pqdict"):
        try:
            importlib.import_module(name)
        except Exception:  # noqa: BLE001
            sys.modules[name] = _Stub(name)
    spec = importlib.util.spec_from_file_location("ref_neural_astar", os.path.join(REF, "__init__.py"), submodule_search_locations=[REF])
    ref = importlib.util.module_from_spec(spec)
    sys.modules["ref_neural_astar"] = ref
    spec.loader.exec_module(ref)

    def params(f):
        return [(p.name, repr(p.default)) for p in inspect.signature(f).parameters.values()]
    diffs, checked = [], 0
    for modname in ("planner.astar", "planner.differentiable_astar", "planner.encoder", "utils.data", "utils.training"):
        r = importlib.import_module("ref_neural_astar." + modname)
        m = importlib.import_module("neural_astar." + modname)
        names = [n for n, o in vars(r).items() if not n.startswith("_") and (inspect.isfunction(o) or inspect.isclass(o))
                 and getattr(o, "__module__", "").startswith("ref_neural_astar")]
        for n in names:
            checked += 1
            ro, mo = getattr(r, n), getattr(m, n, None)
            if mo is None:
                diffs.append(f"{modname}.{n}: missing")
                continue
            a, b = params(ro.__init__ if inspect.isclass(ro) else ro), params(mo.__init__ if inspect.isclass(mo) else mo)
            if b[:len(a)] != a:
                diffs.append(f"{modname}.{n}: parameters {a} vs {b}")
            if inspect.isclass(ro):
                for meth, f in vars(ro).items():
                    if meth.startswith("_") or not inspect.isfunction(f):
                        continue
                    checked += 1
                    g = getattr(mo, meth, None)
                    if g is None:
                        diffs.append(f"{modname}.{n}.{meth}: missing")
                    elif params(g)[:len(params(f))] != params(f):
                        diffs.append(f"{modname}.{n}.{meth}: parameters {params(f)} vs {params(g)}")
    # constructed objects: every public instance attribute / submodule / parameter / buffer of the reference's object exists here, state_dict
    # keys equal (a reference checkpoint loads strict=True)
    ra, ma = importlib.import_module("ref_neural_astar.planner.astar"), importlib.import_module("neural_astar.planner.astar")
    re_, me = importlib.import_module("ref_neural_astar.planner.encoder"), importlib.import_module("neural_astar.planner.encoder")

    def attrs(o):
        return {k for k in vars(o) if not k.startswith("_")} | set(o._modules) | set(o._parameters) | set(o._buffers)
    cases = [("VanillaAstar()", ra.VanillaAstar, ma.VanillaAstar, {}),
             ("NeuralAstar(CNN)", ra.NeuralAstar, ma.NeuralAstar, dict(encoder_arch="CNN")),
             ("NeuralAstar(CNNDownSize, rgb+, 3, const 10, learn_obstacles)", ra.NeuralAstar, ma.NeuralAstar,
              dict(encoder_arch="CNNDownSize", encoder_input="rgb+", encoder_depth=3, const=10.0, learn_obstacles=True)),
             ("DifferentiableAstar()", lambda: ra.VanillaAstar().astar, lambda: ma.VanillaAstar().astar, {}),
             ("encoder.CNN(2, 4, None)", lambda: re_.CNN(2, 4, None), lambda: me.CNN(2, 4, None), {}),
             ("encoder.CNNDownSize(4, 3, 10.0)", lambda: re_.CNNDownSize(4, 3, 10.0), lambda: me.CNNDownSize(4, 3, 10.0), {})]
    for label, rc, mc, kw in cases:
        checked += 1
        r, m = rc(**kw), mc(**kw)
        if attrs(r) - attrs(m):
            diffs.append(f"{label}: instance attributes missing {sorted(attrs(r) - attrs(m))}")
        if list(r.state_dict()) != list(m.state_dict()):
            diffs.append(f"{label}: state_dict keys differ")
    unexpected = [d for d in diffs if d.split(":")[0] not in EXPECTED]
    for d in diffs:
        print(("expected   " if d.split(":")[0] in EXPECTED else "DIFFERENCE ") + d)
    print(json.dumps({"public_names_and_methods_checked": checked, "differences": len(diffs), "unexpected": len(unexpected)}))
    return 1 if unexpected else 0


if __name__ == "__main__":
    sys.exit(main())
