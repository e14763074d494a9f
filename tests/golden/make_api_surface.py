"""Dumps the class / method / argument surface of the reference's compiled-module stubs to
``tests/golden/api_surface.json`` (run in the build container, where ``/root/reference`` exists:
``python tests/golden/make_api_surface.py``).  The output is data - names, argument names, kinds
and defaults - that ``tests/test_api_conformance.py`` compares the shim modules with; the stub
text itself is not kept.

Sources (the three nanobind modules of SURVEY 8(b) + the utility functions this repo covers):
  src/irspack/recommenders/_ials_core.pyi, src/irspack/recommenders/_knn.pyi,
  src/irspack/evaluation/_core_evaluator.pyi, src/irspack/utils/_util_cpp.pyi
"""
import ast
import json
import os
import sys

REF = os.environ.get("IRSPACK_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

MODULES = {
    "irspack_amd.recommenders._ials_core": "src/irspack/recommenders/_ials_core.pyi",
    "irspack_amd.recommenders._knn": "src/irspack/recommenders/_knn.pyi",
    "irspack_amd.evaluation._core_evaluator": "src/irspack/evaluation/_core_evaluator.pyi",
    "irspack_amd.utils": "src/irspack/utils/_util_cpp.pyi",
}
# functions of _util_cpp outside SURVEY 8 (SLIM, splits, the threaded sparse product)
SKIP_FUNCTIONS = {"sparse_mm_threaded", "rowwise_train_test_split_by_ratio",
                  "rowwise_train_test_split_by_fixed_n", "slim_weight_allow_negative",
                  "slim_weight_positive_only"}


def signature(fn: ast.FunctionDef) -> dict:
    a = fn.args
    pos = [x.arg for x in a.posonlyargs]
    args = [x.arg for x in a.args]
    names = pos + args
    defaults = {}
    for name, d in zip(names[len(names) - len(a.defaults):], a.defaults):
        defaults[name] = ast.literal_eval(d)
    if names and names[0] == "self":
        names, pos = names[1:], [p for p in pos if p != "self"]
    return {"args": names, "positional_only": pos, "defaults": defaults}


def surface(path: str) -> dict:
    tree = ast.parse(open(path).read())
    out = {"classes": {}, "functions": {}, "enums": {}}
    for node in tree.body:
        if isinstance(node, ast.ClassDef):
            bases = [ast.unparse(b) for b in node.bases]
            if any(b.endswith("Enum") for b in bases):
                out["enums"][node.name] = {
                    t.targets[0].id: ast.literal_eval(t.value) for t in node.body
                    if isinstance(t, ast.Assign)}
                continue
            methods, props = {}, {}
            for item in node.body:
                if not isinstance(item, ast.FunctionDef):
                    continue
                decos = [ast.unparse(d) for d in item.decorator_list]
                if "property" in decos:
                    props.setdefault(item.name, {"setter": False})
                elif any(d.endswith(".setter") for d in decos):
                    props.setdefault(item.name, {"setter": False})["setter"] = True
                else:  # overloads: one signature per overload
                    methods.setdefault(item.name, []).append(signature(item))
            out["classes"][node.name] = {"methods": methods, "properties": props}
        elif isinstance(node, ast.FunctionDef) and node.name not in SKIP_FUNCTIONS:
            out["functions"][node.name] = signature(node)
    return out


def main() -> None:
    result = {}
    for module, rel in MODULES.items():
        path = os.path.join(REF, rel)
        if not os.path.exists(path):
            sys.exit(f"{path} not found: this script runs where the reference checkout is")
        result[module] = surface(path)
    with open(os.path.join(HERE, "api_surface.json"), "w") as f:
        json.dump(result, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main()
