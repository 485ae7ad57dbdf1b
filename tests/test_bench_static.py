"""bench.py's multi-rank branches cannot run here (no GPU) nor on the one-GPU box (one rank per GPU): a static look at main()
for the class of mistake that only such a branch hides -- a nested helper called at a line above its own definition."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nested_helpers_of_main_are_defined_above_their_first_call():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    nested = {n.name: n for n in main.body if isinstance(n, ast.FunctionDef)}
    assert {"barrier", "init_comm", "gather_outputs"} <= set(nested)

    def calls_outside_nested(node, inside):
        for child in ast.iter_child_nodes(node):
            if isinstance(child, ast.FunctionDef) and child.name in nested:
                continue        # a helper's own body runs when it is called, not where it stands
            if isinstance(child, ast.Call) and isinstance(child.func, ast.Name) and child.func.id in nested:
                yield child.func.id, child.lineno
            yield from calls_outside_nested(child, inside)

    for name, line in calls_outside_nested(main, False):
        assert line > nested[name].end_lineno, "bench.py:%d calls %s() above its definition (line %d)" % (line, name, nested[name].lineno)


def test_bench_compiles_and_names_every_leg():
    src = open(os.path.join(ROOT, "bench.py")).read()
    compile(src, "bench.py", "exec")
    for key in ('"roofline"', '"cpu_baseline"', '"config4_slice"', '"config5_slice"', '"config1"', '"spectrum"', '"after_idle"', '"gather"'):
        assert key in src, key
