"""The Rhai-subset language runtime on the host (no GPU): pfx_script_check evaluates a script with the device-free part of
the host API (print / math / rand / selection flags).  Expected values follow rhai 1.25.1's published semantics — strict
i64 / f64 typing of registered functions, checked integer arithmetic, truncating integer division, value-type arrays,
FloatWrapper formatting — and the reference's own script tests (tests/scripting.rs:30-232)."""
import pytest

import paintfe_amd as P


def out(src, w=64, h=64):
    return P.script_check(src, w, h)


def err(src):
    with pytest.raises(P.PfxError) as e:
        P.script_check(src)
    return e.value


def test_reference_script_tests_that_need_no_pixels():
    assert out("let w = width();\nlet h = height();\nprint_line(`${w}x${h}`);")[-1] == "64x64"    # scripting.rs:31-41
    assert out("let v = clamp(300, 0, 255);\nprint_line(`${v}`);") == ["255"]                       # :201-210
    assert out('print_line("hello world");\nprint_line("second line");') == ["hello world", "second line"]
    assert out('print_line("before: " + has_selection());') == ["before: false"]                     # :329-343
    e = err("let x = ;")                                                                             # :217-223
    assert e.status == -6 and e.line == 1 and str(e)
    e = err("let x = 1 / 0;")                                                                        # :226-232
    assert e.status == -6 and "Division by zero" in str(e)


def test_integer_and_float_semantics():
    assert out("print(7 / 2); print(-7 / 2); print(-7 % 3); print(7.0 / 2); print(2 ** 10); print(2 ** 0.5 > 1.41);") == ["3", "-3", "-1", "3.5", "1024", "true"]
    assert out("print(1 + 2.5); print(3 * 0.5); print(1 == 1.0); print(1 < 1.5); print(\"a\" == 1); print(1 != \"1\");") == ["3.5", "1.5", "true", "true", "false", "true"]
    assert out("print(1 << 4); print(-16 >> 2); print(1 << 64); print(-1 >> 70); print(5 & 3); print(5 | 3); print(5 ^ 3); print(1 << -1);") == \
        ["16", "-4", "0", "-1", "1", "7", "6", "0"]
    assert out("print(0x10 + 0b101 + 0o17 + 1_000);") == ["1036"]
    for src, msg in (("print(9223372036854775807 + 1);", "Addition overflow"), ("print(-9223372036854775807 - 2);", "Subtraction overflow"),
                     ("print(3037000500 * 3037000500);", "Multiplication overflow"), ("print(5 % 0);", "Modulo division by zero"),
                     ("print(2 ** -1);", "negative power"), ("print(10 ** 19);", "Exponential overflow"), ("print(to_int(1e30));", "Integer overflow")):
        assert msg in str(err(src)), src


def test_float_formatting_like_rhai():
    assert out("print(1.5); print(2.0); print(1e20); print(0.1 + 0.2); print(1.0 / 3.0); print(100.0); print(1e-14); print(-0.5); print(0.0); print(12345678.9);") == \
        ["1.5", "2.0", "1e20", "0.30000000000000004", "0.3333333333333333", "100.0", "1e-14", "-0.5", "0.0", "12345678.9"]


def test_operator_precedence_and_short_circuit():
    assert out("print(1 + 2 * 3 ** 2); print(2 ** 3 ** 2); print(-2 ** 2); print(1 + 2 << 3); print(10 - 4 - 3); print(true || false && false);") == \
        ["19", "512", "4", "17", "3", "true"]
    assert out("let x = 0; print(x != 0 && 10 / x > 1); print(x == 0 || 10 / x > 1);") == ["false", "true"]
    assert out("print(!true); print(!(1 > 2));") == ["false", "true"]


def test_variables_blocks_and_control_flow():
    assert out("let x = 0; while x < 5 { x += 1; if x == 3 { continue; } if x == 5 { break; } } print(x);") == ["5"]
    assert out("let t = 0; for i in 0..10 { t += i; } print(t); let u = 0; for i in 0..=10 { u += i; } print(u); for i in range(10, 0, -3) { print(i); }") == \
        ["45", "55", "10", "7", "4", "1"]
    assert out("let n = 0; loop { n += 1; if n > 100 { break; } } print(n); print(if n > 50 { \"big\" } else { \"small\" });") == ["101", "big"]
    assert out("let x = 1; { let x = 2; print(x); } print(x); let y = { let a = 3; a * 2 }; print(y);") == ["2", "1", "6"]
    assert out("let v = if false { 1 }; print(type_of(v)); let z; print(type_of(z));") == ["()", "()"]
    assert "Cannot modify constant" in str(err("const K = 3; K = 4;"))
    assert "Variable not found: nope" in str(err("print(nope);"))
    assert "expecting bool" in str(err("if 1 { }"))


def test_functions_closures_and_arrays():
    assert out("fn fact(n) { if n <= 1 { 1 } else { n * fact(n - 1) } } print(fact(10));") == ["3628800"]
    assert out("fn f(a) { a.push(9); a.len() } let arr = [1, 2]; print(f(arr)); print(arr.len());") == ["3", "2"]  # arguments are copies
    assert out("fn g() { return 5; print(\"unreachable\"); } print(g()); print(later(2)); fn later(x) { x * 21 }") == ["5", "42"]
    assert out("let k = 5; let add = |x| x + k; print(add.call(1)); let mul = |x, y| { let t = x * y; t + 1 }; print(mul.call(3, 4));") == ["6", "13"]
    assert out("let a = [1, 2, 3]; a.push(4); a[0] = 10; a[-1] = 40; print(a); print(a.len()); let s = 0; for v in a { s += v; } print(s);") == \
        ["[10, 2, 3, 40]", "4", "55"]
    assert out("let a = [1, 2]; let b = a; b.push(3); print(a); print(b); print(a + b); print([1, [2, 3]][1][0]);") == ["[1, 2]", "[1, 2, 3]", "[1, 2, 1, 2, 3]", "2"]
    assert "out of bounds" in str(err("let a = [1]; print(a[3]);"))
    assert "Stack overflow" in str(err("fn r(n) { r(n + 1) } r(0);"))
    assert "Function not found: fact (i64, i64)" in str(err("fn fact(n) { n } fact(1, 2);"))


def test_registered_function_typing_is_strict():
    assert "Function not found: apply_blur (i64)" in str(err("apply_blur(2);"))
    assert "Function not found: sqrt (i64)" in str(err("print(sqrt(16));"))
    assert "Function not found: print_line (i64)" in str(err("print_line(5);"))
    assert "Function not found: apply_frobnicate (f64)" in str(err("apply_frobnicate(1.0);"))
    assert out("print(sqrt(16.0)); print(abs(-4)); print(abs(-4.5)); print(max(3, 9)); print(min(2.5, 1.5)); print(floor(3.7)); print((3.7).floor()); "
               "print(to_float(3) * 2.5); print(to_int(3.99)); print(to_int(-3.99)); print(PI()); print(lerp(0.0, 10.0, 0.25)); print(distance(0.0, 0.0, 3.0, 4.0));") == \
        ["4.0", "4", "4.5", "9", "1.5", "3.0", "3.0", "7.5", "3", "-3", "3.141592653589793", "2.5", "5.0"]
    assert out("print(rgb_to_hsl(255, 0, 0)); print(hsl_to_rgb(120.0, 100.0, 50.0)); print(rgb_to_hsl(128, 128, 128));") == \
        ["[0.0, 100.0, 50.0]", "[0, 255, 0]", "[0.0, 0.0, 50.19607843137255]"]
    assert out("let a = rand_int(5, 6); print(a); let f = rand_float(); print(f >= 0.0 && f <= 1.0); print(rand_int(9, 3)); print(rand_float(2.0, 1.0));") == \
        ["5", "true", "9", "2.0"]


def test_strings_and_comments():
    assert out('let n = 3; print(`n=${n}, twice=${n * 2}, s=${"x" + n}`); print("a" + 1 + 2.5 + true); print("tab\\there"); print("q\\"q");') == \
        ["n=3, twice=6, s=x3", "a12.5true", "tab\there", 'q"q']
    assert out("/* nested /* comment */ still comment */ print(1); // trailing\n/// doc comment\nprint(2);") == ["1", "2"]
    assert "not terminated" in str(err('print("abc);'))
    assert "not terminated" in str(err("/* open"))


def test_limits_and_unsupported_constructs():
    assert "Too many operations" in str(err("let i = 0; loop { i += 1; }"))                       # scripting.rs:288
    assert "Size of array too large" in str(err("let a = []; loop { a.push(1); }"))               # :292
    assert "Length of string too large" in str(err('let s = "x"; loop { s += s; }'))              # :291
    for src in ("let m = #{a: 1};", "import \"x\";", "let c = 'c';", "export let x = 1;"):
        assert err(src).status == -5, src
    # image functions need an image
    assert err("apply_invert();").status == -5
    assert err("let p = get_pixel(0, 0);").status == -5


def test_nested_arrays_count_against_the_array_size_limit():
    """Engine::set_max_array_size(10_000), scripting.rs:292: rhai sizes an array as its length plus the sizes of the arrays it holds,
    which also bounds how deep a value can nest"""
    assert "Size of array too large" in str(err("let a = []; for i in 0..20000 { a = [a]; } print(1);"))
    assert "Size of array too large" in str(err("let a = [0]; for i in 0..20000 { let b = [0]; b[0] = a; a = b; } print(1);"))
    assert "Size of array too large" in str(err("let a = []; for i in 0..200 { let b = []; for j in 0..100 { b.push(j); } a.push(b); } print(a.len());"))
    assert "Size of array too large" in str(err("let b = []; for j in 0..6000 { b.push(j); } let c = [b, b]; print(c.len());"))
    assert out("let a = []; for i in 0..90 { let b = []; for j in 0..100 { b.push(j); } a.push(b); } print(a.len());") == ["90"]
    assert out("let a = [0]; for i in 0..4000 { let b = [0]; b[0] = a; a = b; } print(1);") == ["1"]


def test_arrays_are_value_types_however_they_are_shared_internally():
    """copies are lazy inside the runtime (a write clones one level first); what a script sees must be rhai's eager clone-on-assignment"""
    assert out("let a = [1,2,3]; let b = a; b[0] = 9; print(a); print(b);") == ["[1, 2, 3]", "[9, 2, 3]"]
    assert out("let a = [[1,2],[3,4]]; let b = a; b[0][1] = 7; a[1][0] = 0; print(a); print(b);") == ["[[1, 2], [0, 4]]", "[[1, 7], [3, 4]]"]
    assert out("let a = [1]; a.push(a); a[1].push(5); print(a);") == ["[1, [1, 5]]"]
    assert out("fn f(x) { x.push(1); x } let a = []; let b = f(a); print(a.len()); print(b.len());") == ["0", "1"]
    assert out("let a = [1,2]; let c = || a; a.push(3); print(c.call()); print(a);") == ["[1, 2]", "[1, 2, 3]"]
    assert out("let a = [1,2]; for x in a { a.push(x); } print(a);") == ["[1, 2, 1, 2]"]
    assert out("let a = [[0]]; let row = a[0]; row[0] = 5; print(a); print(row); a[0][0] += 2; print(a); print(row);") == ["[[0]]", "[5]", "[[2]]", "[5]"]
    assert out("let a = [1,2]; let b = a + a; b[0] = 7; print(a); print(b); a += [3]; print(a);") == ["[1, 2]", "[7, 2, 1, 2]", "[1, 2, 3]"]
    assert out("let a = [3,1,2]; let b = a; b.reverse(); print(a); print(b); let c = b; c.clear(); print(b.len()); print(c.len());") == \
        ["[3, 1, 2]", "[2, 1, 3]", "3", "0"]


def test_call_levels_times_expression_depth_fit_whatever_stack_the_caller_has():
    """set_max_call_levels(64) x set_max_expr_depths(64, 64), scripting.rs:289-290: the deepest script the sandbox admits needs megabytes of native
    stack; the interpreter brings its own, so a caller on a 256 KB thread stack gets the same answers (and never a fault)"""
    import threading
    deep = "fn f(n) { if n == 0 { 0 } else { 1 + " + "(1 + " * 28 + "f(n - 1)" + ")" * 28 + " } }\nprint(f(%d));"
    got = {}

    def small_stack():
        got["deep"] = out(deep % 63)
        got["level"] = str(err(deep % 70))
        got["closures"] = out("let g = |n, me| if n == 0 { 0 } else { 2 + me.call(n - 1, me) }; print(g.call(60, g));")

    threading.stack_size(256 * 1024)
    try:
        t = threading.Thread(target=small_stack)
        t.start()
        t.join()
    finally:
        threading.stack_size(0)
    assert got["deep"] == [str(63 * 29)] and "Stack overflow" in got["level"] and got["closures"] == ["120"], got


def test_switch_expression():
    src = """
    fn kind(v) {
        switch v {
            0 => "zero",
            1 | 2 | 3 => "small",
            4..10 => "medium",
            10..=99 if v % 2 == 0 => "even tens",
            10..=99 => "odd tens",
            -5 => "minus five",
            "text" => "a string",
            true => "yes",
            2.5 => "float",
            _ => "other"
        }
    }
    for v in [0, 2, 4, 9, 10, 42, 43, 99, 100, -5, "text", true, false, 2.5, 1.0] { print(kind(v)); }
    """
    assert out(src) == ["zero", "small", "medium", "medium", "even tens", "even tens", "odd tens", "odd tens", "other", "minus five", "a string", "yes",
                        "other", "float", "other"]
    # a switch without a matching case and without a default is (); block bodies; usable as a statement without ';'
    assert out("let r = switch 5 { 1 => 2 }; print(r == ()); let n = 0; switch 3 { 3 => { n += 10; n += 1; } _ => { n = -1; } } print(n);") == ["true", "11"]
    assert out("let x = 7; let y = switch x { 7 => { let t = x * 2; t + 1 }, _ => 0 }; print(y);") == ["15"]
    assert "literal" in str(err("let a = 1; switch 1 { a => 2 }"))
    assert "default case" in str(err("switch 1 { _ => 1, 2 => 3 }"))


def test_do_loops():
    assert out("let i = 0; do { i += 1; } while i < 5; print(i); do { i -= 2; } until i <= 0; print(i);") == ["5", "-1"]
    assert out("let i = 0; do { i += 1; if i == 2 { continue; } if i == 4 { break; } } while true; print(i);") == ["4"]
    assert out("let n = 0; do { n += 1; } while false; print(n);") == ["1"]
    assert "Too many operations" in str(err("do { } while true;"))


def test_throw_and_try_catch():
    assert out('try { throw "boom"; } catch (e) { print("caught " + e); }') == ["caught boom"]
    assert out("try { throw 42; } catch (e) { print(e + 1); }") == ["43"]
    assert out("let x = 0; try { x = 10 / x; } catch { x = -1; } print(x);") == ["-1"]
    assert out('try { let a = [1]; a[5]; } catch (e) { print(type_of(e)); }') == ["string"]
    assert out('fn f(n) { if n > 2 { throw `too big: ${n}`; } n } let t = 0; for i in 0..5 { try { t += f(i); } catch (e) { print(e); break; } } print(t);') == \
        ["too big: 3", "3"]
    # nested, re-thrown
    assert out('try { try { throw 1; } catch (e) { throw e + 1; } } catch (e) { print(e); }') == ["2"]
    e = err('let a = 1;\nthrow "fatal " + a;')
    assert e.status == -6 and "Runtime error: fatal 1" in str(e) and e.line == 2
    assert "Runtime error" in str(err("throw;"))
    # the sandbox limits are not catchable
    assert "Too many operations" in str(err("try { loop { } } catch { print(1); }"))
    # break / return pass through a try block
    assert out("fn g() { try { return 5; } catch { return 6; } } print(g()); for i in 0..3 { try { if i == 1 { break; } } catch { } print(i); }") == ["5", "0"]


def test_error_positions():
    e = err("let a = 1;\nlet b = 2;\nlet c = a / (b - 2);\n")
    assert (e.line, e.col) == (3, 11)
    e = err("print(1);\n  frob(2);")
    assert (e.line, e.col) == (2, 3) and "Function not found: frob (i64)" in str(e)
