// pfx_rhai.h — a Rhai-subset language runtime for the script front-end.
//
// The reference embeds the third-party crate rhai 1.25.1 (Cargo.lock; not vendored under the reference tree) and
// registers its host API in src/ops/scripting.rs:284-1482.  This runtime restates the part of the published language
// that effect scripts use — `let`/`const`, i64/f64/bool/string/array values with Rhai's strict typing and checked
// integer arithmetic, `if`/`while`/`loop`/`do .. while|until`/`for .. in`, `switch` expressions (literal, `|` alternatives,
// integer ranges, `if` guards, `_`), `throw` / `try .. catch (e)` (a caught runtime error binds its message string — Rhai
// binds an object map, which this subset does not have), `fn` definitions, closures, method-call syntax, back-tick string
// interpolation — as a tree-walking interpreter for the *control plane* of a script, plus a compiler that lowers
// per-pixel closures (`map_channels`, `for_each_pixel`, `for_region`) to a register bytecode executed by one GPU
// kernel (k_script.hip).  Pixel data never goes through the interpreter except for the scalar get_pixel/set_pixel API.
//
// Operation budget (scripting.rs:288 set_max_operations(50_000_000)): the interpreter counts like Rhai.  A per-pixel closure
// compiled for the GPU gets, per pixel, max(4096, (50 M - operations so far) / pixels of the call) bytecode steps
// (pfx_script_host.cpp): a runaway loop ends with 'Too many operations' after a bounded launch, a closure whose per-pixel cost
// exceeds both the floor and its share of the budget fails as in the reference.  Remaining divergence, on purpose: a cheap
// closure over a large image (7 steps x 33 Mpx) exhausts the reference's global budget but runs here.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace rhai {

struct Node;
using NodeP = std::shared_ptr<Node>;

struct Closure;
struct Value {
    enum T : uint8_t { Unit, Int, Float, Bool, Str, Array, Fn, Range } t = Unit;
    int64_t i = 0;       // Int; Range start
    int64_t j = 0;       // Range end (exclusive after normalisation)
    double f = 0.0;
    bool b = false;
    std::shared_ptr<std::string> s;
    std::shared_ptr<std::vector<Value>> a;
    std::shared_ptr<Closure> fn;

    static Value unit() { return Value(); }
    static Value from_int(int64_t v) { Value x; x.t = Int; x.i = v; return x; }
    static Value from_float(double v) { Value x; x.t = Float; x.f = v; return x; }
    static Value from_bool(bool v) { Value x; x.t = Bool; x.b = v; return x; }
    static Value from_str(const std::string& v) { Value x; x.t = Str; x.s = std::make_shared<std::string>(v); return x; }
    static Value from_array(std::vector<Value> v) { Value x; x.t = Array; x.a = std::make_shared<std::vector<Value>>(std::move(v)); return x; }
    Value copy() const; // arrays are value types in Rhai: assignment / argument passing clones (lazily: see own())
    std::vector<Value>& own(); // Array only: the element vector, made private to this value before a write
    std::string to_string() const;
    const char* type_name() const; // as Rhai prints it in "Function not found" messages
};

struct Closure {
    std::vector<std::string> params;
    NodeP body;
    std::map<std::string, Value> captured; // by value at creation (Rhai shares captured variables; effect scripts only read them)
    std::string fn_name;                   // non-empty: pointer to a script-defined fn (`Fn("name")`)
};

enum class NK : uint8_t {
    IntLit, FloatLit, BoolLit, StrLit, Interp, ArrayLit, Var, Unary, Binary, And, Or, Call, Index, ClosureLit, If, Block, Let, Assign,
    While, Loop, For, Break, Continue, Return, FnDef, ExprStmt, RangeLit,
    DoWhile,  // kids: body, condition; flag = `until`
    Throw,    // kids: [value]
    Try,      // kids: body, handler; text = the catch variable (may be empty)
    Switch,   // kids: scrutinee, arms...
    Arm       // kids: patterns (ival of them), [guard if flag], body; text == "_" for the default arm
};

struct Node {
    NK k;
    int line = 0, col = 0;
    std::string text;        // identifier / operator / string
    int64_t ival = 0;
    double fval = 0.0;
    bool flag = false;       // ExprStmt: terminated by ';' | Let: const | RangeLit: inclusive | Call: method-call syntax
    std::vector<NodeP> kids;
    std::vector<std::string> params; // ClosureLit / FnDef
};

struct Error {
    std::string msg;
    int line = 0, col = 0;
    int status = 0; // PFX_ERR_SCRIPT by default, PFX_ERR_UNSUPPORTED for constructs outside the subset
};

// ---- device bytecode for per-pixel closures (executed by k_script.hip) ------------------------------------------------
// 64-bit registers per lane (i64 or f64 bit patterns; bools are 0/1).  Registers 0..5 are preloaded with the closure's
// parameters (map_channels: r,g,b,a in 0..3; for_each_pixel / for_region: x,y,r,g,b,a in 0..5).
enum BcOp : uint16_t {
    BC_LOADK = 0, BC_MOV,
    BC_IADD, BC_ISUB, BC_IMUL, BC_IDIV, BC_IMOD, BC_INEG, BC_IPOW, BC_IAND, BC_IOR, BC_IXOR, BC_ISHL, BC_ISHR, BC_IABS, BC_IMIN, BC_IMAX, BC_ICLAMP,
    BC_ISIGN,
    BC_IEQ, BC_INE, BC_ILT, BC_ILE, BC_IGT, BC_IGE,
    BC_FADD, BC_FSUB, BC_FMUL, BC_FDIV, BC_FMOD, BC_FNEG, BC_FPOW, BC_FABS, BC_FMIN, BC_FMAX, BC_FCLAMP, BC_FFLOOR, BC_FCEIL, BC_FROUND, BC_FSQRT,
    BC_FSIN, BC_FCOS, BC_FTAN, BC_FATAN2, BC_FEXP, BC_FLN, BC_FLERP, BC_FDIST,
    BC_FEQ, BC_FNE, BC_FLT, BC_FLE, BC_FGT, BC_FGE,
    BC_I2F, BC_F2I, BC_NOT,
    BC_JMP, BC_JZ, BC_JNZ,
    BC_GETCH,  // dst = channel c (0..3) of the source image at (reg a, reg b), 0 outside
    BC_ISSEL,  // dst = is_selected(reg a, reg b)
    BC_RET_ARR, // a = first of 4 consecutive registers, b = bit mask of elements that are integers (others keep the old channel)
    BC_RET_UNIT,
    BC_ERR,    // a = error code (BcErr)
    BC_COUNT
};
enum BcErr : int32_t { BCE_NONE = 0, BCE_ADD_OVERFLOW, BCE_SUB_OVERFLOW, BCE_MUL_OVERFLOW, BCE_DIV_ZERO, BCE_DIV_OVERFLOW, BCE_MOD_ZERO, BCE_NEG_OVERFLOW,
             BCE_POW_OVERFLOW, BCE_POW_NEGATIVE, BCE_F2I_RANGE, BCE_TOO_MANY_OPS, BCE_SHIFT, BCE_ABS_OVERFLOW };
struct BcIns { uint16_t op, dst, a, b, c, line; };
struct BcProgram {
    std::vector<BcIns> code;
    std::vector<uint64_t> consts;
    int n_regs = 6;
    int n_params = 4; // 4: map_channels, 6: for_each_pixel / for_region
    int n_pre = 0;    // code[0 .. n_pre) are LOADKs into registers nothing else writes: a lane runs them once, every pixel starts at n_pre (hoist_constants)
};

// ---- host interface -----------------------------------------------------------------------------------------------------
class Interp;
struct Host {
    virtual ~Host() {}
    // Registered function lookup.  Returns 0 = no function of that name, 1 = name exists but no overload for these argument
    // types, 2 = called (out / err filled; err.msg non-empty on failure).
    virtual int call(Interp& in, const std::string& name, std::vector<Value>& args, Value& out, Error& err) = 0;
};

class Interp {
public:
    explicit Interp(Host* host) : host_(host) {}
    bool run(const char* source, Error& err);      // parse + execute, on a thread of its own with a stack sized for the sandbox's depth limits
    bool run_here(const char* source, Error& err); // the same on the caller's thread (stack use is checked either way)
    std::function<void()> on_run_thread;          // called first on the thread that evaluates (the script host binds its HIP device there)
    uint64_t ops() const { return ops_; }
    uint64_t max_ops = 50000000ull;               // Engine::set_max_operations, scripting.rs:288 (lowered by the hardening harness only)
    std::vector<std::string> console;
    // compile a closure to device bytecode (n_params 4 or 6; img_w / img_h resolve width() / height())
    bool compile_closure(const Closure& c, int n_params, int64_t img_w, int64_t img_h, BcProgram& out, Error& err);
    // host-side call of a closure / fn pointer (used by nothing on the pixel path; available for completeness)
    bool call_closure(const Closure& c, std::vector<Value>& args, Value& out, Error& err);
    static const char* bc_error_text(int code);

private:
    friend struct Eval;
    Host* host_;
    uint64_t ops_ = 0;
    std::map<std::string, NodeP> fns_; // script-defined functions, keyed "name/arity"
};

} // namespace rhai
