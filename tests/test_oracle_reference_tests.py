"""The two known-answer tests the REFERENCE itself holds for a function on this path: `addition` and `fib` in
rusteria/src/lib.rs:274-296 (`Rusteria::execute_string`), which run `Execution::execute` (rusteria/src/node/execution.rs:109-752,
SURVEY.md §8 row F3) over the NodeOp program its compiler emits and assert the value left on the stack.  They are the only
assertions anywhere in the reference that exercise code of the §8 path, so they are the only reference-held pins the oracle can
have; the rasterizer proper has none and stays "parity unpinned".

The reference's front end (scanner / parser / compiler) is out of scope and not restated; the NodeOp lists below are what
rusteria/src/compile.rs emits for the two source strings, derived by reading it:
  * a top-level `let a = <e>` is a global (parser.rs:137-142) -> <e>, StoreGlobal(i)                          (compile.rs:437-452)
  * a number literal                                          -> Push(Value::broadcast(f))                     (compile.rs:689-690)
  * a variable                                                -> LoadLocal(i) inside a function whose locals hold it, else LoadGlobal(i)
                                                                                                               (compile.rs:654-662)
  * `l <op> r`                                                -> l, r, Add / Sub / Le ...                      (compile.rs:745-792)
  * `if c {t} else {e}`                                       -> c, If(t, Some(e))  (the branches are compiled first, the condition
                                                                 is emitted in front of the If)                 (compile.rs:1011-1038)
  * `return e`                                                -> e, Return                                     (compile.rs:999-1007)
  * `f(args)` of a user function                              -> args, FunctionCall(n_args, n_locals, index)   (compile.rs:887-908)
  * an expression statement emits its expression only (compile.rs:412-419); optimize() is empty (optimize.rs:17-19);
    `execute` pops the result off the stack (lib.rs:132-139).
The harness runs the main body as function 0 of the program set, so the user function `fib` (index 0 in the reference's
`program.user_functions`) is function 1 here; nothing else differs."""
import pytest

from tests.test_oracle_vm import color_of


def test_reference_addition(oracle):
    # lib.rs:274-279: "let a = 2; a + 2;" == 4.0
    body = [("Push", 2.0), ("StoreGlobal", 0), ("LoadGlobal", 0), ("Push", 2.0), "Add"]
    assert color_of(oracle, body, globals=1) == (4.0, 4.0, 4.0)


FIB = [("LoadLocal", 0), ("Push", 1.0), "Le",
       ("If", [("LoadLocal", 0), "Return"],
        [("LoadLocal", 0), ("Push", 1.0), "Sub", ("FunctionCall", 1, 1, 1),
         ("LoadLocal", 0), ("Push", 2.0), "Sub", ("FunctionCall", 1, 1, 1), "Add", "Return"])]


def test_reference_fib(oracle):
    # lib.rs:281-296: fib(27) == 196418.0 (832 039 calls, 27 deep: the reference recurses on the Rust stack without a bound)
    assert color_of(oracle, [("Push", 27.0), ("FunctionCall", 1, 1, 1)], functions=[FIB]) == (196418.0, 196418.0, 196418.0)


@pytest.mark.parametrize("n,want", [(0, 0.0), (1, 1.0), (2, 1.0), (7, 13.0), (20, 6765.0)])
def test_reference_fib_small(oracle, n, want):
    assert color_of(oracle, [("Push", float(n)), ("FunctionCall", 1, 1, 1)], functions=[FIB])[0] == want
