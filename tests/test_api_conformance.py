"""The shim modules against the reference's stubs (``_ials_core.pyi``, ``_knn.pyi``,
``_core_evaluator.pyi``, the covered functions of ``_util_cpp.pyi``), through the surface dump
``tests/golden/api_surface.json`` (made by ``tests/golden/make_api_surface.py`` where the
reference checkout is).  Every class, method, property (and setter), enum member, module-level
function, argument NAME, ORDER and DEFAULT of the reference must exist here; the shims may take
more keyword arguments (``device=...``) behind them, never fewer or renamed ones.
"""
import importlib
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SURFACE = json.load(open(os.path.join(HERE, "golden", "api_surface.json")))


def params_of(fn):
    ps = [p for p in inspect.signature(fn).parameters.values() if p.name != "self"]
    return ps


def check_signature(where, fn, ref_sigs):
    """``fn`` must accept every overload of the reference: same leading argument names in the
    same order (positional-only ones are free to be named), same defaults; extras need defaults."""
    ps = params_of(fn)
    if any(p.kind == p.VAR_POSITIONAL for p in ps):
        pytest.fail(f"{where}: *args hides the argument names")
    longest = max(ref_sigs, key=lambda s: len(s["args"]))
    names = [p.name for p in ps if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    for i, a in enumerate(longest["args"]):
        assert i < len(names), f"{where}: missing argument {a!r}"
        if a not in longest["positional_only"]:
            assert names[i] == a, f"{where}: argument {i} is {names[i]!r}, the reference names it {a!r}"
    by_name = {p.name: p for p in ps}
    for a, d in longest["defaults"].items():
        p = by_name[a]
        assert p.default is not inspect.Parameter.empty and p.default == d, \
            f"{where}: default of {a!r} is {p.default!r}, the reference has {d!r}"
    for p in ps[len(longest["args"]):]:  # what the shim adds must be optional
        if p.kind != p.VAR_KEYWORD:
            assert p.default is not inspect.Parameter.empty, \
                f"{where}: extra argument {p.name!r} needs a default"
    if len(ref_sigs) > 1:  # overloads: the arguments beyond the shortest one must be optional
        n_min = min(len(s["args"]) for s in ref_sigs)
        for p in ps[n_min:len(longest["args"])]:
            assert p.default is not inspect.Parameter.empty, \
                f"{where}: {p.name!r} is absent from one overload and needs a default"


@pytest.mark.parametrize("module_name", sorted(SURFACE))
def test_module_surface(module_name):
    mod = importlib.import_module(module_name)
    ref = SURFACE[module_name]
    for ename, members in ref["enums"].items():
        e = getattr(mod, ename)
        assert {m.name: m.value for m in e} == members, ename
    for fname, sig in ref["functions"].items():
        assert hasattr(mod, fname), f"{module_name}.{fname} is missing"
        check_signature(f"{module_name}.{fname}", getattr(mod, fname), [sig])
    for cname, c in ref["classes"].items():
        assert hasattr(mod, cname), f"{module_name}.{cname} is missing"
        cls = getattr(mod, cname)
        for pname, prop in c["properties"].items():
            attr = inspect.getattr_static(cls, pname)
            assert isinstance(attr, property), f"{cname}.{pname} must be a property"
            if prop["setter"]:
                assert attr.fset is not None, f"{cname}.{pname} needs a setter"
        for mname, sigs in c["methods"].items():
            assert hasattr(cls, mname), f"{cname}.{mname} is missing"
            check_signature(f"{cname}.{mname}", getattr(cls, mname), sigs)


def test_module_level_enum_aliases():
    # _ials_core.pyi exports the enum members at module level too (nanobind export_values)
    m = importlib.import_module("irspack_amd.recommenders._ials_core")
    assert m.ORIGINAL is m.LossType.ORIGINAL and m.CHOLESKY is m.SolverType.CHOLESKY
    assert m.CG is m.SolverType.CG and m.IALSPP is m.SolverType.IALSPP


def test_caller_classes_cover_the_reference_methods():
    """Public methods of the reference's Python callers on the hot path (names only: the
    orchestration methods - tune, from_config, learn_with_optimizer - are out of scope)."""
    from irspack_amd.evaluation import Evaluator, EvaluatorWithColdUser
    from irspack_amd.recommenders.ials import IALSRecommender, IALSTrainer

    for cls, names in [
        (Evaluator, ["get_target_score", "get_score", "get_scores", "get_score_from_score_matrix",
                     "get_scores_from_score_matrix", "get_score_from_score_chunks",
                     "get_scores_from_score_chunks"]),
        (EvaluatorWithColdUser, ["get_score", "get_scores", "get_score_from_score_matrix",
                                 "get_score_from_score_chunks"]),
        (IALSTrainer, ["load_state", "save_state", "compute_loss", "run_epoch", "user_scores",
                       "transform_user", "transform_item", "transform_user_feature",
                       "transform_item_feature"]),
        (IALSRecommender, ["get_score", "get_score_block", "get_score_cold_user",
                           "get_score_cold_user_with_item_features", "get_user_embedding",
                           "get_item_embedding", "get_score_from_user_embedding",
                           "get_score_from_item_embedding", "compute_user_embedding",
                           "compute_user_embedding_from_features",
                           "get_score_cold_user_from_features", "compute_item_embedding",
                           "compute_item_embedding_from_features", "get_score_from_item_features",
                           "get_score_remove_seen", "get_score_remove_seen_block",
                           "get_score_cold_user_remove_seen"]),
    ]:
        for n in names:
            assert callable(getattr(cls, n, None)), f"{cls.__name__}.{n} is missing"
    sig = inspect.signature(EvaluatorWithColdUser.__init__)
    assert sig.parameters["mb_size"].default == 1024 and "cold_item_features" in sig.parameters
    assert inspect.signature(Evaluator.__init__).parameters["mb_size"].default == 128
