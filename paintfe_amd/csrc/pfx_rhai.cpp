// pfx_rhai.cpp — lexer, parser, tree-walking interpreter and per-pixel-closure compiler of the Rhai subset (see pfx_rhai.h).
//
// Language rules follow the published behaviour of rhai 1.25.1 (the version pinned by the reference's Cargo.lock):
// operator precedence table (|| | ^ : 30, && & : 60, == != : 90, < <= > >= : 130, .. ..= : 140, + - : 150, * / % : 180,
// ** : 190 right-assoc, << >> : 210), checked i64 arithmetic ("Addition overflow", "Division by zero", ...), i64/f64 mixing in the
// built-in arithmetic and comparison operators but NOT in registered-function dispatch ("Function not found: apply_blur (i64)"),
// arrays as value types, `name(args)` == `arg0.name(rest)`, statements vs trailing block expressions, nested block comments.
// Call sites that anchor the behaviour: src/ops/scripting.rs:284-317 (engine limits), tests/scripting.rs (reference tests).
#include "pfx_rhai.h"

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <set>

#include <pthread.h>

namespace rhai {

namespace {
constexpr int ST_SCRIPT = -6, ST_UNSUPPORTED = -5; // PFX_ERR_SCRIPT / PFX_ERR_UNSUPPORTED (include/pfx.h)
constexpr int MAX_CALL_LEVELS = 64;                // scripting.rs:289
constexpr size_t MAX_STRING = 10000, MAX_ARRAY = 10000; // scripting.rs:291-292

// ---- native stack ------------------------------------------------------------------------------------------------------
// The sandbox allows 64 call levels of expressions nested 64 deep (scripting.rs:289-290): some 4 000 evaluator frames, more native stack
// than a caller's thread may have (a 1 MB thread stack ends at ~30 x 63).  Interp::run therefore evaluates on a thread of its own with
// RUN_STACK_BYTES of stack, and every evaluator / parser frame checks its address against the low end of whichever stack it is on, so
// that exhaustion is the script error "Stack overflow" (what Rhai reports for its call-level limit), never a fault.
#if !defined(__has_feature)
#define __has_feature(x) 0
#endif
#if defined(__SANITIZE_ADDRESS__) || __has_feature(address_sanitizer)
constexpr size_t RUN_STACK_BYTES = 128u << 20, STACK_MARGIN = 1u << 20; // instrumented frames are larger; the sanitizer clears shadow memory for the whole stack of every thread, so not more than this
#else
constexpr size_t RUN_STACK_BYTES = 64u << 20, STACK_MARGIN = 256u << 10;
#endif
thread_local uintptr_t t_stack_floor = 0; // frames below this address fail; 0 = unknown (no check)

uintptr_t this_threads_stack_floor()
{
    pthread_attr_t at;
    if (pthread_getattr_np(pthread_self(), &at) != 0) return 0;
    void* lo = nullptr;
    size_t size = 0;
    const int rc = pthread_attr_getstack(&at, &lo, &size);
    pthread_attr_destroy(&at);
    if (rc != 0 || size <= 2 * STACK_MARGIN) return 0;
    return (uintptr_t)lo + STACK_MARGIN;
}
struct StackFloorScope {
    uintptr_t saved;
    StackFloorScope() : saved(t_stack_floor) { t_stack_floor = this_threads_stack_floor(); }
    ~StackFloorScope() { t_stack_floor = saved; }
};
inline bool stack_left() { return (uintptr_t)__builtin_frame_address(0) > t_stack_floor; }

struct Throw { Error e; std::shared_ptr<Value> value; }; // value: what a script's `throw` statement raised
[[noreturn]] void fail(const std::string& m, int line, int col, int status = ST_SCRIPT) { throw Throw{{m, line, col, status}}; }
[[noreturn]] void fail(const std::string& m, const Node& n, int status = ST_SCRIPT) { fail(m, n.line, n.col, status); }

// Engine::set_max_array_size (scripting.rs:292) counts the elements of nested arrays too (rhai calc_data_sizes): an array's size is its length
// plus the sizes of the arrays it holds.  Counting stops at the limit, which also bounds how deep a value can nest (copies and destructors recurse).
size_t array_total(const Value& v, size_t budget)
{
    size_t n = v.a->size();
    for (const Value& e : *v.a) {
        if (n > budget) break;
        if (e.t == Value::Array) n += array_total(e, budget - n);
    }
    return n;
}
void check_array(const Value& v, const Node& at)
{
    if (v.t == Value::Array && array_total(v, MAX_ARRAY) > MAX_ARRAY) fail("Size of array too large", at);
}

// ---- float formatting (rhai FloatWrapper Display) ---------------------------------------------------------------------
void shortest_digits(double v, std::string& digits, int& exp10)
{
    char buf[64];
    for (int p = 1; p <= 17; ++p) {
        std::snprintf(buf, sizeof buf, "%.*e", p - 1, v);
        if (std::strtod(buf, nullptr) == v) break;
    }
    const char* e = std::strchr(buf, 'e');
    digits.clear();
    for (const char* c = buf; c < e; ++c) if (std::isdigit((unsigned char)*c)) digits += *c;
    exp10 = std::atoi(e + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
}
std::string fmt_float(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    const double a = std::fabs(v);
    if (a == 0.0) return "0.0";
    std::string d;
    int e;
    shortest_digits(a, d, e);
    std::string out = v < 0 ? "-" : "";
    if (a > 10000000000000.0 || a < 0.0000000000001) { // `{:e}`
        out += d[0];
        if (d.size() > 1) { out += '.'; out += d.substr(1); }
        out += "e" + std::to_string(e);
        return out;
    }
    if (e >= 0) {
        if ((int)d.size() <= e + 1) { out += d + std::string(e + 1 - d.size(), '0') + ".0"; }
        else out += d.substr(0, e + 1) + "." + d.substr(e + 1);
    } else out += "0." + std::string(-e - 1, '0') + d;
    return out;
}
} // namespace

// Arrays are value types.  A copy shares the element vector until one side writes: every write goes through own() (Eval::lvalue for `a[i] = ..`,
// the mutating methods' receiver), which takes a private one-level clone first if the vector is shared.  Observable behaviour is that of an eager deep
// copy; the cost of passing a nested array around no longer grows with its size (a loop of `a = [a]` was quadratic).
Value Value::copy() const { return *this; }
std::vector<Value>& Value::own()
{
    if (a.use_count() > 1) a = std::make_shared<std::vector<Value>>(*a);
    return *a;
}
std::string Value::to_string() const
{
    switch (t) {
    case Unit: return "";
    case Int: return std::to_string(i);
    case Float: return fmt_float(f);
    case Bool: return b ? "true" : "false";
    case Str: return s ? *s : std::string();
    case Array: {
        std::string o = "[";
        for (size_t k = 0; a && k < a->size(); ++k) {
            if (k) o += ", ";
            const Value& e = (*a)[k];
            o += e.t == Str ? "\"" + e.to_string() + "\"" : e.to_string();
        }
        return o + "]";
    }
    case Fn: return "Fn(" + (fn && !fn->fn_name.empty() ? fn->fn_name : std::string("anon")) + ")";
    case Range: return std::to_string(i) + ".." + std::to_string(j);
    }
    return "";
}
const char* Value::type_name() const
{
    switch (t) {
    case Unit: return "()";
    case Int: return "i64";
    case Float: return "f64";
    case Bool: return "bool";
    case Str: return "&str | ImmutableString | String";
    case Array: return "array";
    case Fn: return "Fn";
    case Range: return "range";
    }
    return "?";
}

// =========================================================================================================== lexer
namespace {
enum class TT : uint8_t { End, Ident, Int, Float, Str, Interp, Punct };
struct Tok {
    TT t = TT::End;
    std::string text;
    int64_t i = 0;
    double f = 0.0;
    int line = 1, col = 1;
    std::vector<std::pair<bool, std::string>> parts; // Interp: (is_expr, text)
};

class Lexer {
public:
    Lexer(const char* s, int line0 = 1, int col0 = 1) : p_(s), line_(line0), col_(col0) {}
    Tok next()
    {
        skip();
        Tok o;
        o.line = line_;
        o.col = col_;
        const char c = *p_;
        if (c == '\0') return o;
        if (std::isalpha((unsigned char)c) || c == '_') {
            const char* s = p_;
            while (std::isalnum((unsigned char)*p_) || *p_ == '_') adv();
            o.t = TT::Ident;
            o.text.assign(s, p_);
            return o;
        }
        if (std::isdigit((unsigned char)c)) return number(o);
        if (c == '"') return string_lit(o);
        if (c == '`') return interp_lit(o);
        if (c == '\'') fail("character literals are outside the supported Rhai subset", o.line, o.col, ST_UNSUPPORTED);
        static const char* multi[] = {"**=", "<<=", ">>=", "..=", "==", "!=", "<=", ">=", "&&", "||", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=",
                                      "**", "<<", ">>", "..", "=>", "::", "??", "?.", "#{"};
        for (const char* m : multi) {
            const size_t n = std::strlen(m);
            if (std::strncmp(p_, m, n) == 0) {
                for (size_t k = 0; k < n; ++k) adv();
                o.t = TT::Punct;
                o.text = m;
                return o;
            }
        }
        adv();
        o.t = TT::Punct;
        o.text = std::string(1, c);
        if (!std::strchr("+-*/%=<>!&|^(){}[],;.:", c)) fail(std::string("Unknown character '") + c + "'", o.line, o.col);
        return o;
    }

private:
    void adv()
    {
        if (*p_ == '\n') { ++line_; col_ = 1; } else ++col_;
        ++p_;
    }
    void skip()
    {
        for (;;) {
            while (std::isspace((unsigned char)*p_)) adv();
            if (p_[0] == '/' && p_[1] == '/') { while (*p_ && *p_ != '\n') adv(); continue; }
            if (p_[0] == '/' && p_[1] == '*') {
                const int l = line_, c = col_;
                int depth = 0; // block comments nest
                do {
                    if (p_[0] == '/' && p_[1] == '*') { ++depth; adv(); adv(); }
                    else if (p_[0] == '*' && p_[1] == '/') { --depth; adv(); adv(); }
                    else adv();
                } while (*p_ && depth > 0);
                if (depth > 0) fail("Open block comment is not terminated", l, c);
                continue;
            }
            break;
        }
    }
    Tok number(Tok o)
    {
        const char* s = p_;
        if (p_[0] == '0' && (p_[1] == 'x' || p_[1] == 'X' || p_[1] == 'o' || p_[1] == 'O' || p_[1] == 'b' || p_[1] == 'B')) {
            const int base = (p_[1] == 'x' || p_[1] == 'X') ? 16 : ((p_[1] == 'o' || p_[1] == 'O') ? 8 : 2);
            adv(); adv();
            std::string d;
            while (std::isalnum((unsigned char)*p_) || *p_ == '_') { if (*p_ != '_') d += *p_; adv(); }
            char* end = nullptr;
            errno = 0;
            const unsigned long long v = std::strtoull(d.c_str(), &end, base);
            if (d.empty() || *end || errno) fail("Malformed number: " + std::string(s, p_), o.line, o.col);
            o.t = TT::Int;
            o.i = (int64_t)v;
            return o;
        }
        bool is_float = false;
        while (std::isdigit((unsigned char)*p_) || *p_ == '_') adv();
        if (*p_ == '.' && std::isdigit((unsigned char)p_[1])) { is_float = true; adv(); while (std::isdigit((unsigned char)*p_) || *p_ == '_') adv(); }
        else if (*p_ == '.' && p_[1] != '.' && !std::isalpha((unsigned char)p_[1]) && p_[1] != '_') { is_float = true; adv(); } // `4.` is a float
        if (*p_ == 'e' || *p_ == 'E') {
            const char* q = p_ + 1;
            if (*q == '+' || *q == '-') ++q;
            if (std::isdigit((unsigned char)*q)) { is_float = true; while (p_ < q) adv(); while (std::isdigit((unsigned char)*p_)) adv(); }
        }
        std::string num(s, p_);
        std::string clean;
        for (char ch : num) if (ch != '_') clean += ch;
        if (is_float) { o.t = TT::Float; o.f = std::strtod(clean.c_str(), nullptr); }
        else {
            errno = 0;
            const long long v = std::strtoll(clean.c_str(), nullptr, 10);
            if (errno) fail("Malformed number: " + num, o.line, o.col);
            o.t = TT::Int;
            o.i = v;
        }
        o.text = clean;
        return o;
    }
    char escape()
    {
        adv(); // backslash
        const char e = *p_;
        if (!e) return '\\';
        adv();
        switch (e) {
        case 'n': return '\n';
        case 't': return '\t';
        case 'r': return '\r';
        case '0': return '\0';
        default: return e;
        }
    }
    Tok string_lit(Tok o)
    {
        adv();
        std::string s;
        while (*p_ && *p_ != '"') {
            if (*p_ == '\n') break; // a normal string literal cannot span lines
            if (*p_ == '\\') s += escape();
            else { s += *p_; adv(); }
        }
        if (*p_ != '"') fail("Open string is not terminated", o.line, o.col);
        adv();
        if (s.size() > MAX_STRING) fail("Length of string too large", o.line, o.col);
        o.t = TT::Str;
        o.text = s;
        return o;
    }
    Tok interp_lit(Tok o)
    {
        adv();
        std::string cur;
        while (*p_ && *p_ != '`') {
            if (p_[0] == '`' && p_[1] == '`') { cur += '`'; adv(); adv(); continue; }
            if (p_[0] == '$' && p_[1] == '{') {
                o.parts.push_back({false, cur});
                cur.clear();
                adv(); adv();
                int depth = 1;
                std::string ex;
                while (*p_ && depth > 0) {
                    if (*p_ == '{') ++depth;
                    else if (*p_ == '}') { if (--depth == 0) break; }
                    ex += *p_;
                    adv();
                }
                if (*p_ != '}') fail("Open string is not terminated", o.line, o.col);
                adv();
                o.parts.push_back({true, ex});
                continue;
            }
            cur += *p_;
            adv();
        }
        if (*p_ != '`') fail("Open string is not terminated", o.line, o.col);
        adv();
        o.parts.push_back({false, cur});
        o.t = TT::Interp;
        return o;
    }
    const char* p_;
    int line_, col_;
};

// =========================================================================================================== parser
NodeP mk(NK k, const Tok& t) { auto n = std::make_shared<Node>(); n->k = k; n->line = t.line; n->col = t.col; return n; }

class Parser {
public:
    explicit Parser(const char* src, int line0 = 1, int col0 = 1, int depth0 = 0) : lx_(src, line0, col0), depth_(depth0) { cur_ = lx_.next(); }
    NodeP program()
    {
        auto b = mk(NK::Block, cur_);
        while (cur_.t != TT::End) {
            if (is_p(";")) { eat(); continue; }
            b->kids.push_back(statement());
        }
        return b;
    }
    NodeP single_expr()
    {
        NodeP e = expr(0);
        if (cur_.t != TT::End) fail("Unexpected '" + cur_.text + "' in string interpolation", cur_.line, cur_.col);
        return e;
    }

private:
    Lexer lx_;
    Tok cur_;
    int depth_ = 0;
    struct DepthGuard {
        Parser& p;
        explicit DepthGuard(Parser& q, const Tok& t) : p(q) { if (++p.depth_ > 64 || !stack_left()) fail("Expression exceeds maximum complexity", t.line, t.col); } // scripting.rs:290
        ~DepthGuard() { --p.depth_; }
    };
    void eat() { cur_ = lx_.next(); }
    bool is_p(const char* s) const { return cur_.t == TT::Punct && cur_.text == s; }
    bool is_kw(const char* s) const { return cur_.t == TT::Ident && cur_.text == s; }
    void expect_p(const char* s, const std::string& what)
    {
        if (!is_p(s)) fail("Expecting '" + std::string(s) + "' " + what, cur_.line, cur_.col);
        eat();
    }
    std::string ident(const std::string& what)
    {
        if (cur_.t != TT::Ident) fail("Expecting " + what, cur_.line, cur_.col);
        static const std::set<std::string> reserved = {"let", "const", "if", "else", "while", "loop", "for", "in", "fn", "return", "break", "continue", "true",
                                                       "false", "switch", "do", "until", "throw", "try", "catch", "import", "export", "as", "private", "this"};
        if (reserved.count(cur_.text)) fail("'" + cur_.text + "' is a reserved keyword", cur_.line, cur_.col);
        std::string s = cur_.text;
        eat();
        return s;
    }
    NodeP block()
    {
        DepthGuard g(*this, cur_);   // nested blocks count against the same limit as nested expressions (Engine::set_max_expr_depths, scripting.rs:290)
        auto b = mk(NK::Block, cur_);
        expect_p("{", "to start a statement block");
        while (!is_p("}")) {
            if (cur_.t == TT::End) fail("Expecting '}' to terminate this block", cur_.line, cur_.col);
            if (is_p(";")) { eat(); continue; }
            b->kids.push_back(statement());
        }
        eat();
        return b;
    }
    static bool block_like(const NodeP& e)
    {
        return e->k == NK::If || e->k == NK::While || e->k == NK::Loop || e->k == NK::For || e->k == NK::Block || e->k == NK::Switch || e->k == NK::Try;
    }
    void end_stmt(const NodeP& st, bool needs_semi)
    {
        if (is_p(";")) { eat(); st->flag = true; return; }
        if (needs_semi && !is_p("}") && cur_.t != TT::End) fail("Expecting ';' to terminate this statement", cur_.line, cur_.col);
    }
    NodeP statement()
    {
        const Tok t0 = cur_;
        if (is_kw("let") || is_kw("const")) {
            auto n = mk(NK::Let, t0);
            const bool is_const = is_kw("const");
            eat();
            n->text = ident("a variable name");
            if (is_p("=")) { eat(); n->kids.push_back(expr(0)); }
            else if (is_const) fail("Expecting '=' to assign a value to this constant", cur_.line, cur_.col);
            n->ival = is_const ? 1 : 0;
            auto st = mk(NK::ExprStmt, t0);
            st->kids.push_back(n);
            end_stmt(st, true);
            st->flag = true;
            return st;
        }
        if (is_kw("fn") || is_kw("private")) {
            if (is_kw("private")) eat();
            if (!is_kw("fn")) fail("Expecting 'fn'", cur_.line, cur_.col);
            eat();
            auto n = mk(NK::FnDef, t0);
            n->text = ident("a function name");
            expect_p("(", "to start the parameters list of function '" + n->text + "'");
            while (!is_p(")")) {
                n->params.push_back(ident("a parameter name"));
                if (is_p(",")) eat();
                else if (!is_p(")")) fail("Expecting ',' to separate the parameters of function '" + n->text + "'", cur_.line, cur_.col);
            }
            eat();
            n->kids.push_back(block());
            return n;
        }
        if (is_kw("while")) {
            eat();
            auto n = mk(NK::While, t0);
            n->kids.push_back(expr(0));
            n->kids.push_back(block());
            if (is_p(";")) eat();
            return wrap_stmt(n, t0);
        }
        if (is_kw("loop")) {
            eat();
            auto n = mk(NK::Loop, t0);
            n->kids.push_back(block());
            if (is_p(";")) eat();
            return wrap_stmt(n, t0);
        }
        if (is_kw("for")) {
            eat();
            auto n = mk(NK::For, t0);
            if (is_p("(")) fail("'for (value, counter) in' is outside the supported Rhai subset", cur_.line, cur_.col, ST_UNSUPPORTED);
            n->text = ident("a variable name");
            if (!is_kw("in")) fail("Expecting 'in' after the loop variable", cur_.line, cur_.col);
            eat();
            n->kids.push_back(expr(0));
            n->kids.push_back(block());
            if (is_p(";")) eat();
            return wrap_stmt(n, t0);
        }
        if (is_kw("break") || is_kw("continue")) {
            auto n = mk(is_kw("break") ? NK::Break : NK::Continue, t0);
            eat();
            auto st = wrap_stmt(n, t0);
            end_stmt(st, true);
            st->flag = true;
            return st;
        }
        if (is_kw("return")) {
            eat();
            auto n = mk(NK::Return, t0);
            if (!is_p(";") && !is_p("}") && cur_.t != TT::End) n->kids.push_back(expr(0));
            auto st = wrap_stmt(n, t0);
            end_stmt(st, true);
            st->flag = true;
            return st;
        }
        if (is_kw("do")) { // do { ... } while cond;  /  do { ... } until cond;
            eat();
            auto n = mk(NK::DoWhile, t0);
            n->kids.push_back(block());
            if (!is_kw("while") && !is_kw("until")) fail("Expecting 'while' or 'until' after the body of this do loop", cur_.line, cur_.col);
            n->flag = is_kw("until");
            eat();
            n->kids.push_back(expr(0));
            auto st = wrap_stmt(n, t0);
            end_stmt(st, true);
            st->flag = true;
            return st;
        }
        if (is_kw("throw")) {
            eat();
            auto n = mk(NK::Throw, t0);
            if (!is_p(";") && !is_p("}") && cur_.t != TT::End) n->kids.push_back(expr(0));
            auto st = wrap_stmt(n, t0);
            end_stmt(st, true);
            st->flag = true;
            return st;
        }
        if (is_kw("try")) {
            eat();
            auto n = mk(NK::Try, t0);
            n->kids.push_back(block());
            if (!is_kw("catch")) fail("Expecting 'catch' after the body of this try statement", cur_.line, cur_.col);
            eat();
            if (is_p("(")) {
                eat();
                n->text = ident("a variable name for the caught error");
                expect_p(")", "to close the catch variable");
            }
            n->kids.push_back(block());
            if (is_p(";")) eat();
            return wrap_stmt(n, t0);
        }
        for (const char* kw : {"import", "export"})
            if (is_kw(kw)) fail("Rhai statement '" + cur_.text + "' is outside the supported subset", cur_.line, cur_.col, ST_UNSUPPORTED);
        // expression or assignment
        NodeP e = expr(0);
        static const char* assign_ops[] = {"=", "+=", "-=", "*=", "/=", "%=", "**=", "<<=", ">>=", "&=", "|=", "^="};
        for (const char* op : assign_ops)
            if (is_p(op)) {
                if (e->k != NK::Var && e->k != NK::Index) fail("Cannot assign to this expression", cur_.line, cur_.col);
                auto n = mk(NK::Assign, cur_);
                n->text = op;
                eat();
                n->kids.push_back(e);
                n->kids.push_back(expr(0));
                auto st = wrap_stmt(n, t0);
                end_stmt(st, true);
                st->flag = true;
                return st;
            }
        auto st = wrap_stmt(e, t0);
        end_stmt(st, !block_like(e));
        return st;
    }
    NodeP wrap_stmt(const NodeP& e, const Tok& t0)
    {
        auto st = mk(NK::ExprStmt, t0);
        st->kids.push_back(e);
        return st;
    }
    static int prec(const Tok& t)
    {
        if (t.t != TT::Punct) return -1;
        const std::string& s = t.text;
        if (s == "||" || s == "|" || s == "^") return 30;
        if (s == "&&" || s == "&") return 60;
        if (s == "==" || s == "!=") return 90;
        if (s == "<" || s == "<=" || s == ">" || s == ">=") return 130;
        if (s == ".." || s == "..=") return 140;
        if (s == "+" || s == "-") return 150;
        if (s == "*" || s == "/" || s == "%") return 180;
        if (s == "**") return 190;
        if (s == "<<" || s == ">>") return 210;
        return -1;
    }
    NodeP expr(int min_prec)
    {
        DepthGuard g(*this, cur_);
        NodeP lhs = unary();
        for (;;) {
            const int p = prec(cur_);
            if (p < 0 || p < min_prec) return lhs;
            const Tok op = cur_;
            eat();
            NodeP rhs = expr(op.text == "**" ? p : p + 1); // ** binds to the right
            NodeP n;
            if (op.text == "&&") n = mk(NK::And, op);
            else if (op.text == "||") n = mk(NK::Or, op);
            else if (op.text == ".." || op.text == "..=") { n = mk(NK::RangeLit, op); n->flag = op.text == "..="; }
            else { n = mk(NK::Binary, op); n->text = op.text; }
            n->kids = {lhs, rhs};
            lhs = n;
        }
    }
    NodeP unary()
    {
        if (is_p("-") || is_p("!") || is_p("+")) {
            const Tok op = cur_;
            eat();
            DepthGuard g(*this, op);
            NodeP v = unary();
            if (op.text == "+") return v;
            if (op.text == "-" && v->k == NK::IntLit && v->text != "neg") { v->ival = (int64_t)(0ull - (uint64_t)v->ival); v->text = "neg"; return v; }
            if (op.text == "-" && v->k == NK::FloatLit) { v->fval = -v->fval; return v; }
            auto n = mk(NK::Unary, op);
            n->text = op.text;
            n->kids.push_back(v);
            return n;
        }
        return postfix(primary());
    }
    std::vector<NodeP> call_args(const std::string& fname)
    {
        std::vector<NodeP> a;
        eat(); // (
        while (!is_p(")")) {
            if (cur_.t == TT::End) fail("Expecting ')' to close the parameters list of function call '" + fname + "'", cur_.line, cur_.col);
            a.push_back(expr(0));
            if (is_p(",")) eat();
            else if (cur_.t == TT::End) fail("Expecting ')' to close the parameters list of function call '" + fname + "'", cur_.line, cur_.col);
            else if (!is_p(")")) fail("Expecting ',' to separate the parameters of function call '" + fname + "'", cur_.line, cur_.col);
        }
        eat();
        return a;
    }
    NodeP postfix(NodeP e)
    {
        for (;;) {
            if (is_p("[")) {
                auto n = mk(NK::Index, cur_);
                eat();
                n->kids.push_back(e);          // not `= {e, expr(0)}`: g++ 11 leaks the constructed elements of a braced list when a later one throws (GCC PR 66139)
                n->kids.push_back(expr(0));
                expect_p("]", "to close this index expression");
                e = n;
            } else if (is_p(".")) {
                const Tok dot = cur_;
                eat();
                if (cur_.t != TT::Ident) fail("Expecting a method or property name after '.'", cur_.line, cur_.col);
                auto n = mk(NK::Call, cur_);
                n->text = cur_.text;
                n->flag = true; // method-call syntax
                eat();
                if (is_p("(")) {
                    std::vector<NodeP> a = call_args(n->text);
                    n->kids.push_back(e);
                    for (auto& x : a) n->kids.push_back(x);
                } else { // property getter: only `.len` style getters of the standard library
                    n->kids.push_back(e);
                    n->ival = 1;
                }
                (void)dot;
                e = n;
            } else return e;
        }
    }
    NodeP closure_lit(const Tok& t0, bool no_params)
    {
        auto n = mk(NK::ClosureLit, t0);
        if (!no_params) {
            while (!is_p("|")) {
                n->params.push_back(ident("a closure parameter name"));
                if (is_p(",")) eat();
                else if (!is_p("|")) fail("Expecting ',' to separate the parameters of this closure", cur_.line, cur_.col);
            }
            eat();
        }
        NodeP body = is_p("{") ? block() : expr(0);
        n->kids.push_back(body);
        return n;
    }
    NodeP if_expr()
    {
        auto n = mk(NK::If, cur_);
        eat();
        n->kids.push_back(expr(0));
        n->kids.push_back(block());
        if (is_kw("else")) {
            eat();
            if (is_kw("if")) n->kids.push_back(if_expr());
            else n->kids.push_back(block());
        }
        return n;
    }
    // switch value { pattern | pattern if guard => expr-or-block, lo..hi => ..., _ => ... }
    NodeP switch_expr()
    {
        auto n = mk(NK::Switch, cur_);
        eat();
        n->kids.push_back(expr(0));
        expect_p("{", "to start the cases of this switch expression");
        bool seen_default = false;
        while (!is_p("}")) {
            if (cur_.t == TT::End) fail("Expecting '}' to terminate this switch expression", cur_.line, cur_.col);
            auto arm = mk(NK::Arm, cur_);
            if (cur_.t == TT::Ident && cur_.text == "_") {
                if (seen_default) fail("Duplicated default case in this switch expression", cur_.line, cur_.col);
                seen_default = true;
                arm->text = "_";
                eat();
            } else {
                if (seen_default) fail("The default case must be the last case of this switch expression", cur_.line, cur_.col);
                for (;;) {
                    NodeP pat = expr(35); // above '|' (which separates alternatives here), so ranges and negative literals parse whole
                    const Node* lit = pat.get();
                    if (lit->k == NK::Unary && lit->text == "-" && !lit->kids.empty()) lit = lit->kids[0].get();
                    const bool ok = lit->k == NK::IntLit || lit->k == NK::FloatLit || lit->k == NK::BoolLit || lit->k == NK::StrLit ||
                                    (pat->k == NK::RangeLit && pat->kids.size() == 2);
                    if (!ok) fail("A switch case must be a literal value or an integer range", pat->line, pat->col);
                    arm->kids.push_back(pat);
                    if (is_p("|")) { eat(); continue; }
                    break;
                }
            }
            arm->ival = (int64_t)arm->kids.size();
            if (is_kw("if")) { eat(); arm->flag = true; arm->kids.push_back(expr(0)); }
            expect_p("=>", "after the case of this switch expression");
            const bool blk = is_p("{");
            arm->kids.push_back(blk ? block() : expr(0));
            n->kids.push_back(arm);
            if (is_p(",")) eat();
            else if (!is_p("}") && !blk) fail("Expecting ',' to separate the cases of this switch expression", cur_.line, cur_.col);
        }
        eat();
        return n;
    }
    NodeP primary()
    {
        const Tok t = cur_;
        switch (t.t) {
        case TT::Int: { eat(); auto n = mk(NK::IntLit, t); n->ival = t.i; return n; }
        case TT::Float: { eat(); auto n = mk(NK::FloatLit, t); n->fval = t.f; return n; }
        case TT::Str: { eat(); auto n = mk(NK::StrLit, t); n->text = t.text; return n; }
        case TT::Interp: {
            eat();
            auto n = mk(NK::Interp, t);
            for (const auto& part : t.parts) {
                if (!part.first) { auto s = mk(NK::StrLit, t); s->text = part.second; n->kids.push_back(s); }
                else {   // an interpolated expression nests like a parenthesised one: the sub-parser inherits the depth (and pays one level)
                    DepthGuard g(*this, t);
                    Parser sub(part.second.c_str(), t.line, t.col, depth_);
                    n->kids.push_back(sub.single_expr());
                }
            }
            return n;
        }
        case TT::Ident: {
            if (t.text == "true" || t.text == "false") { eat(); auto n = mk(NK::BoolLit, t); n->ival = t.text == "true"; return n; }
            if (t.text == "if") return if_expr();
            if (t.text == "switch") return switch_expr();
            for (const char* kw : {"while", "loop", "for", "do", "fn", "let", "const", "return", "break", "continue", "throw", "try", "this"})
                if (t.text == kw) fail("'" + t.text + "' is not allowed in this expression position (outside the supported subset)", t.line, t.col, ST_UNSUPPORTED);
            eat();
            if (is_p("(")) {
                auto n = mk(NK::Call, t);
                n->text = t.text;
                n->kids = call_args(t.text);
                return n;
            }
            if (is_p("::")) fail("module paths are outside the supported Rhai subset", cur_.line, cur_.col, ST_UNSUPPORTED);
            auto n = mk(NK::Var, t);
            n->text = t.text;
            return n;
        }
        case TT::Punct:
            if (t.text == "(") {
                eat();
                if (is_p(")")) { eat(); auto u = mk(NK::Block, t); return u; } // `()` = unit
                NodeP e = expr(0);
                expect_p(")", "to close this expression");
                return e;
            }
            if (t.text == "[") {
                eat();
                auto n = mk(NK::ArrayLit, t);
                while (!is_p("]")) {
                    if (cur_.t == TT::End) fail("Expecting ']' to close this array literal", cur_.line, cur_.col);
                    n->kids.push_back(expr(0));
                    if (is_p(",")) eat();
                    else if (!is_p("]")) fail("Expecting ',' to separate the items of this array literal", cur_.line, cur_.col);
                }
                eat();
                if (n->kids.size() > MAX_ARRAY) fail("Size of array too large", t.line, t.col);
                return n;
            }
            if (t.text == "{") return block();
            if (t.text == "|") { eat(); return closure_lit(t, false); }
            if (t.text == "||") { eat(); return closure_lit(t, true); }
            if (t.text == "#{") fail("object maps are outside the supported Rhai subset", t.line, t.col, ST_UNSUPPORTED);
            if (t.text == ";" || t.text == ")" || t.text == "}" || t.text == "]" || t.text == "=" || t.text == ",")
                fail("Unexpected '" + t.text + "'", t.line, t.col);
            fail("Unexpected '" + t.text + "'", t.line, t.col);
        case TT::End: fail("Script is incomplete", t.line, t.col);
        }
        fail("Unexpected token", t.line, t.col);
    }
};

// =========================================================================================================== operators
inline bool add_ov(int64_t a, int64_t b, int64_t& r) { return __builtin_add_overflow(a, b, &r); }
inline bool sub_ov(int64_t a, int64_t b, int64_t& r) { return __builtin_sub_overflow(a, b, &r); }
inline bool mul_ov(int64_t a, int64_t b, int64_t& r) { return __builtin_mul_overflow(a, b, &r); }

bool int_pow(int64_t base, int64_t e, int64_t& out)
{
    int64_t r = 1;
    while (e > 0) {
        if (e & 1) { if (mul_ov(r, base, r)) return false; }
        e >>= 1;
        if (e > 0 && mul_ov(base, base, base)) return false;
    }
    out = r;
    return true;
}

Value binary_op(const std::string& op, const Value& a, const Value& b, const Node& at)
{
    using V = Value;
    if (a.t == V::Int && b.t == V::Int) {
        const int64_t x = a.i, y = b.i;
        int64_t r = 0;
        const std::string ctx = ": " + std::to_string(x) + " " + op + " " + std::to_string(y);
        if (op == "+") { if (add_ov(x, y, r)) fail("Addition overflow" + ctx, at); return V::from_int(r); }
        if (op == "-") { if (sub_ov(x, y, r)) fail("Subtraction overflow" + ctx, at); return V::from_int(r); }
        if (op == "*") { if (mul_ov(x, y, r)) fail("Multiplication overflow" + ctx, at); return V::from_int(r); }
        if (op == "/") {
            if (y == 0) fail("Division by zero" + ctx, at);
            if (x == std::numeric_limits<int64_t>::min() && y == -1) fail("Division overflow" + ctx, at);
            return V::from_int(x / y);
        }
        if (op == "%") {
            if (y == 0) fail("Modulo division by zero" + ctx, at);
            if (x == std::numeric_limits<int64_t>::min() && y == -1) fail("Modulo division overflow" + ctx, at);
            return V::from_int(x % y);
        }
        if (op == "**") {
            if (y < 0) fail("Integer raised to a negative power" + ctx, at);
            if (!int_pow(x, y, r)) fail("Exponential overflow" + ctx, at);
            return V::from_int(r);
        }
        if (op == "&") return V::from_int(x & y);
        if (op == "|") return V::from_int(x | y);
        if (op == "^") return V::from_int(x ^ y);
        if (op == "<<" || op == ">>") {
            bool left = op == "<<";
            int64_t n = y;
            if (n < 0) { left = !left; n = (n == std::numeric_limits<int64_t>::min()) ? 64 : -n; }
            if (left) return V::from_int(n >= 64 ? 0 : (int64_t)((uint64_t)x << n));
            return V::from_int(n >= 64 ? (x < 0 ? -1 : 0) : (x >> n));
        }
        if (op == "==") return V::from_bool(x == y);
        if (op == "!=") return V::from_bool(x != y);
        if (op == "<") return V::from_bool(x < y);
        if (op == "<=") return V::from_bool(x <= y);
        if (op == ">") return V::from_bool(x > y);
        if (op == ">=") return V::from_bool(x >= y);
    }
    const bool an = a.t == V::Int || a.t == V::Float, bn = b.t == V::Int || b.t == V::Float;
    if (an && bn) { // at least one float: built-in mixed arithmetic promotes the integer
        const double x = a.t == V::Int ? (double)a.i : a.f, y = b.t == V::Int ? (double)b.i : b.f;
        if (op == "+") return V::from_float(x + y);
        if (op == "-") return V::from_float(x - y);
        if (op == "*") return V::from_float(x * y);
        if (op == "/") return V::from_float(x / y);
        if (op == "%") return V::from_float(std::fmod(x, y));
        if (op == "**") return V::from_float(b.t == V::Int ? std::pow(x, (double)b.i) : std::pow(x, y));
        if (op == "==") return V::from_bool(x == y);
        if (op == "!=") return V::from_bool(x != y);
        if (op == "<") return V::from_bool(x < y);
        if (op == "<=") return V::from_bool(x <= y);
        if (op == ">") return V::from_bool(x > y);
        if (op == ">=") return V::from_bool(x >= y);
    }
    if (a.t == V::Bool && b.t == V::Bool) {
        if (op == "==") return V::from_bool(a.b == b.b);
        if (op == "!=") return V::from_bool(a.b != b.b);
        if (op == "&") return V::from_bool(a.b && b.b);
        if (op == "|") return V::from_bool(a.b || b.b);
        if (op == "^") return V::from_bool(a.b != b.b);
    }
    if (a.t == V::Str || b.t == V::Str) {
        if (op == "+" && a.t != V::Array && b.t != V::Array && a.t != V::Fn && b.t != V::Fn) {
            std::string s = a.to_string() + b.to_string();
            if (s.size() > MAX_STRING) fail("Length of string too large", at);
            return V::from_str(s);
        }
        if (a.t == V::Str && b.t == V::Str) {
            const int c = a.s->compare(*b.s);
            if (op == "==") return V::from_bool(c == 0);
            if (op == "!=") return V::from_bool(c != 0);
            if (op == "<") return V::from_bool(c < 0);
            if (op == "<=") return V::from_bool(c <= 0);
            if (op == ">") return V::from_bool(c > 0);
            if (op == ">=") return V::from_bool(c >= 0);
        }
    }
    if (a.t == V::Array && b.t == V::Array && op == "+") {
        std::vector<Value> r = *a.a;
        for (const Value& e : *b.a) r.push_back(e.copy());
        V sum = V::from_array(std::move(r));
        check_array(sum, at);
        return sum;
    }
    if (a.t == V::Unit && b.t == V::Unit && (op == "==" || op == "!=")) return V::from_bool(op == "==");
    if (op == "==") return V::from_bool(false); // different types never compare equal
    if (op == "!=") return V::from_bool(true);
    fail(std::string("Function not found: ") + op + " (" + a.type_name() + ", " + b.type_name() + ")", at);
}

bool values_equal(const Value& a, const Value& b)
{
    if (a.t != b.t) {
        if ((a.t == Value::Int && b.t == Value::Float) || (a.t == Value::Float && b.t == Value::Int))
            return (a.t == Value::Int ? (double)a.i : a.f) == (b.t == Value::Int ? (double)b.i : b.f);
        return false;
    }
    switch (a.t) {
    case Value::Unit: return true;
    case Value::Int: return a.i == b.i;
    case Value::Float: return a.f == b.f;
    case Value::Bool: return a.b == b.b;
    case Value::Str: return *a.s == *b.s;
    case Value::Array:
        if (a.a->size() != b.a->size()) return false;
        for (size_t k = 0; k < a.a->size(); ++k) if (!values_equal((*a.a)[k], (*b.a)[k])) return false;
        return true;
    default: return false;
    }
}

void collect_idents(const NodeP& n, std::set<std::string>& out)
{
    if (!n) return;
    if (n->k == NK::Var) out.insert(n->text);
    for (const auto& k : n->kids) collect_idents(k, out);
}

std::string signature(const std::string& name, const std::vector<Value>& args)
{
    std::string s = name + " (";
    for (size_t k = 0; k < args.size(); ++k) { if (k) s += ", "; s += args[k].type_name(); }
    return s + ")";
}
} // namespace

// =========================================================================================================== evaluator
struct Eval {
    Interp& in;
    struct Var { Value v; bool is_const; };
    std::vector<std::map<std::string, Var>> scopes;
    int call_depth = 0;
    struct Brk {};
    struct Cont {};
    struct Ret { Value v; };

    explicit Eval(Interp& i) : in(i) { scopes.emplace_back(); }

    void tick(const Node& n)
    {
        if (++in.ops_ > in.max_ops) fail("Too many operations", n);
    }
    Var* find(const std::string& name)
    {
        for (size_t k = scopes.size(); k-- > 0;) {
            auto it = scopes[k].find(name);
            if (it != scopes[k].end()) return &it->second;
        }
        return nullptr;
    }
    struct ScopeGuard {
        Eval& e;
        explicit ScopeGuard(Eval& x) : e(x) { e.scopes.emplace_back(); }
        ~ScopeGuard() { e.scopes.pop_back(); }
    };

    Value block(const Node& b, bool new_scope = true)
    {
        std::unique_ptr<ScopeGuard> g;
        if (new_scope) g.reset(new ScopeGuard(*this));
        Value last;
        for (size_t k = 0; k < b.kids.size(); ++k) {
            const Node& st = *b.kids[k];
            if (st.k == NK::FnDef) continue; // hoisted
            Value v = eval(st);
            last = (k + 1 == b.kids.size() && st.k == NK::ExprStmt && !st.flag) ? v : Value();
        }
        return last;
    }

    bool truthy(const Value& v, const Node& at)
    {
        if (v.t != Value::Bool) fail(std::string("Data type incorrect: ") + v.type_name() + " (expecting bool)", at);
        return v.b;
    }

    Value* lvalue(const Node& target, bool& is_const)
    {
        if (target.k == NK::Var) {
            Var* v = find(target.text);
            if (!v) fail("Variable not found: " + target.text, target);
            is_const = v->is_const;
            return &v->v;
        }
        if (target.k == NK::Index) {
            Value* base = lvalue(*target.kids[0], is_const);
            Value idx = eval(*target.kids[1]);
            if (base->t != Value::Array) fail(std::string("Indexing is not supported for type ") + base->type_name(), target, base->t == Value::Str ? ST_UNSUPPORTED : ST_SCRIPT);
            if (idx.t != Value::Int) fail(std::string("Data type incorrect: ") + idx.type_name() + " (expecting i64)", target);
            const int64_t n = (int64_t)base->a->size();
            int64_t i = idx.i < 0 ? n + idx.i : idx.i;
            if (i < 0 || i >= n) fail("Array index " + std::to_string(idx.i) + " out of bounds: only " + std::to_string(n) + " elements in the array", target);
            return &base->own()[(size_t)i];
        }
        fail("Cannot assign to this expression", target);
    }

    Value call_script_fn(const Node& def, std::vector<Value>& args, const Node& at)
    {
        if (++call_depth > MAX_CALL_LEVELS) { --call_depth; fail("Stack overflow", at); }
        std::vector<std::map<std::string, Var>> saved;
        saved.swap(scopes); // script functions are pure: no access to the caller's variables
        scopes.emplace_back();
        for (size_t k = 0; k < def.params.size(); ++k) scopes.back()[def.params[k]] = {args[k].copy(), false};
        Value out;
        try {
            out = block(*def.kids[0], false);
        } catch (Ret& r) {
            out = r.v;
        } catch (...) {
            scopes.swap(saved);
            --call_depth;
            throw;
        }
        scopes.swap(saved);
        --call_depth;
        return out;
    }

    Value call_closure(const Closure& c, std::vector<Value>& args, const Node& at)
    {
        if (!c.fn_name.empty()) {
            auto it = in.fns_.find(c.fn_name + "/" + std::to_string(args.size()));
            if (it == in.fns_.end()) fail("Function not found: " + signature(c.fn_name, args), at);
            return call_script_fn(*it->second, args, at);
        }
        if (args.size() != c.params.size())
            fail("Function not found: " + signature("anon", args), at);
        if (++call_depth > MAX_CALL_LEVELS) { --call_depth; fail("Stack overflow", at); }
        std::vector<std::map<std::string, Var>> saved;
        saved.swap(scopes);
        scopes.emplace_back();
        for (const auto& kv : c.captured) scopes.back()[kv.first] = {kv.second, false};
        for (size_t k = 0; k < c.params.size(); ++k) scopes.back()[c.params[k]] = {args[k].copy(), false};
        Value out;
        try {
            out = c.body->k == NK::Block ? block(*c.body, false) : eval(*c.body);
        } catch (Ret& r) {
            out = r.v;
        } catch (...) {
            scopes.swap(saved);
            --call_depth;
            throw;
        }
        scopes.swap(saved);
        --call_depth;
        return out;
    }

    // standard-library functions that effect scripts use (the registered host API lives behind Host::call)
    bool builtin(const std::string& name, std::vector<Value>& a, Value& out, const Node& at, Value* recv_lvalue)
    {
        using V = Value;
        const size_t n = a.size();
        auto is = [&](size_t k, V::T t) { return k < n && a[k].t == t; };
        if (name == "print" && n == 1) { in.console.push_back(a[0].to_string()); return true; }
        if (name == "debug" && n == 1) { in.console.push_back(a[0].t == V::Str ? "\"" + a[0].to_string() + "\"" : a[0].to_string()); return true; }
        if (name == "to_string" && n == 1) { out = V::from_str(a[0].to_string()); return true; }
        if (name == "type_of" && n == 1) {
            static const char* names[] = {"()", "i64", "f64", "bool", "string", "array", "Fn", "range"};
            out = V::from_str(names[a[0].t]);
            return true;
        }
        if (name == "to_float" && n == 1 && (is(0, V::Int) || is(0, V::Float))) { out = V::from_float(is(0, V::Int) ? (double)a[0].i : a[0].f); return true; }
        if (name == "to_int" && n == 1 && is(0, V::Int)) { out = a[0]; return true; }
        if (name == "to_int" && n == 1 && is(0, V::Float)) {
            const double f = a[0].f;
            if (!(f > -9223372036854775809.0 && f < 9223372036854775808.0)) fail("Integer overflow: to_int(" + fmt_float(f) + ")", at);
            out = V::from_int((int64_t)f);
            return true;
        }
        if (name == "sign" && n == 1 && is(0, V::Int)) { out = V::from_int(a[0].i > 0 ? 1 : (a[0].i < 0 ? -1 : 0)); return true; }
        if (name == "sign" && n == 1 && is(0, V::Float)) { out = V::from_int(a[0].f > 0 ? 1 : (a[0].f < 0 ? -1 : 0)); return true; }
        if ((name == "is_odd" || name == "is_even" || name == "is_zero") && n == 1 && is(0, V::Int)) {
            out = V::from_bool(name == "is_zero" ? a[0].i == 0 : ((a[0].i & 1) != 0) == (name == "is_odd"));
            return true;
        }
        if (n == 1 && is(0, V::Float)) { // f64 methods of the standard library not re-registered by the host
            const double x = a[0].f;
            if (name == "exp") { out = V::from_float(std::exp(x)); return true; }
            if (name == "ln") { out = V::from_float(std::log(x)); return true; }
            if (name == "log10" || name == "log") { out = V::from_float(std::log10(x)); return true; }
            if (name == "asin") { out = V::from_float(std::asin(x)); return true; }
            if (name == "acos") { out = V::from_float(std::acos(x)); return true; }
            if (name == "atan") { out = V::from_float(std::atan(x)); return true; }
            if (name == "sinh") { out = V::from_float(std::sinh(x)); return true; }
            if (name == "cosh") { out = V::from_float(std::cosh(x)); return true; }
            if (name == "tanh") { out = V::from_float(std::tanh(x)); return true; }
            if (name == "fraction") { out = V::from_float(x - std::trunc(x)); return true; }
            if (name == "int") { out = V::from_float(std::trunc(x)); return true; }
            if (name == "is_nan") { out = V::from_bool(std::isnan(x)); return true; }
            if (name == "to_degrees") { out = V::from_float(x * (180.0 / 3.14159265358979323846)); return true; }
            if (name == "to_radians") { out = V::from_float(x * (3.14159265358979323846 / 180.0)); return true; }
        }
        if (name == "range" && (n == 2 || n == 3) && is(0, V::Int) && is(1, V::Int) && (n == 2 || is(2, V::Int))) {
            out = V();
            out.t = V::Range;
            out.i = a[0].i;
            out.j = a[1].i;
            out.f = n == 3 ? (double)a[2].i : 1.0; // step
            if (n == 3 && a[2].i == 0) fail("range: step is zero", at);
            return true;
        }
        if (name == "Fn" && n == 1 && is(0, V::Str)) {
            out = V();
            out.t = V::Fn;
            out.fn = std::make_shared<Closure>();
            out.fn->fn_name = *a[0].s;
            return true;
        }
        if (name == "len" && n == 1 && is(0, V::Array)) { out = V::from_int((int64_t)a[0].a->size()); return true; }
        if (name == "len" && n == 1 && is(0, V::Str)) { out = V::from_int((int64_t)a[0].s->size()); return true; }
        if (name == "is_empty" && n == 1 && is(0, V::Array)) { out = V::from_bool(a[0].a->empty()); return true; }
        if (name == "contains" && n == 2 && is(0, V::Array)) {
            bool f = false;
            for (const Value& e : *a[0].a) f = f || values_equal(e, a[1]);
            out = V::from_bool(f);
            return true;
        }
        if (name == "contains" && n == 2 && is(0, V::Str) && is(1, V::Str)) { out = V::from_bool(a[0].s->find(*a[1].s) != std::string::npos); return true; }
        if ((name == "to_upper" || name == "to_lower") && n == 1 && is(0, V::Str)) {
            std::string s = *a[0].s;
            for (char& c : s) c = name == "to_upper" ? (char)std::toupper((unsigned char)c) : (char)std::tolower((unsigned char)c);
            out = V::from_str(s);
            return true;
        }
        // array mutators act on the receiver variable
        if (n >= 1 && is(0, V::Array) && recv_lvalue && recv_lvalue->t == V::Array) {
            std::vector<Value>& arr = recv_lvalue->own();
            if (name == "push" && n == 2) {
                if (arr.size() >= MAX_ARRAY) fail("Size of array too large", at);
                arr.push_back(a[1].copy());
                if (a[1].t == V::Array) check_array(*recv_lvalue, at);
                return true;
            }
            if (name == "pop" && n == 1) { if (!arr.empty()) { out = arr.back(); arr.pop_back(); } return true; }
            if (name == "shift" && n == 1) { if (!arr.empty()) { out = arr.front(); arr.erase(arr.begin()); } return true; }
            if (name == "clear" && n == 1) { arr.clear(); return true; }
            if (name == "reverse" && n == 1) { std::reverse(arr.begin(), arr.end()); return true; }
            if (name == "truncate" && n == 2 && is(1, V::Int)) { if (a[1].i >= 0 && (size_t)a[1].i < arr.size()) arr.resize((size_t)a[1].i); return true; }
            if (name == "insert" && n == 3 && is(1, V::Int)) {
                int64_t i = a[1].i < 0 ? std::max<int64_t>(0, (int64_t)arr.size() + a[1].i) : std::min<int64_t>(a[1].i, (int64_t)arr.size());
                if (arr.size() >= MAX_ARRAY) fail("Size of array too large", at);
                arr.insert(arr.begin() + i, a[2].copy());
                if (a[2].t == V::Array) check_array(*recv_lvalue, at);
                return true;
            }
            if (name == "remove" && n == 2 && is(1, V::Int)) {
                int64_t i = a[1].i < 0 ? (int64_t)arr.size() + a[1].i : a[1].i;
                if (i >= 0 && i < (int64_t)arr.size()) { out = arr[(size_t)i]; arr.erase(arr.begin() + i); }
                return true;
            }
            if (name == "append" && n == 2 && is(1, V::Array)) {
                for (const Value& e : *a[1].a) arr.push_back(e.copy());
                check_array(*recv_lvalue, at);
                return true;
            }
        }
        if (name == "call" && n >= 1 && is(0, V::Fn)) {
            std::vector<Value> rest(a.begin() + 1, a.end());
            out = call_closure(*a[0].fn, rest, at);
            return true;
        }
        return false;
    }

    Value call(const Node& n)
    {
        std::vector<Value> args;
        args.reserve(n.kids.size());
        for (const auto& k : n.kids) args.push_back(eval(*k));
        // 1. script-defined functions
        auto it = in.fns_.find(n.text + "/" + std::to_string(args.size()));
        if (it != in.fns_.end()) return call_script_fn(*it->second, args, n);
        // 2. a variable holding a closure, called by name
        if (!n.flag) {
            if (Var* v = find(n.text)) {
                if (v->v.t == Value::Fn) { Closure c = *v->v.fn; return call_closure(c, args, n); }
            }
        }
        // 3. standard library
        Value out;
        Value* recv = nullptr;
        bool is_const = false;
        if (n.flag && !n.kids.empty() && (n.kids[0]->k == NK::Var || n.kids[0]->k == NK::Index) && !args.empty() && args[0].t == Value::Array) {
            recv = lvalue(*n.kids[0], is_const);
            if (is_const) recv = nullptr;
        }
        if (builtin(n.text, args, out, n, recv)) return out;
        // 4. the host's registered API
        Error herr;
        const int r = in.host_ ? in.host_->call(in, n.text, args, out, herr) : 0;
        if (r == 2) {
            if (!herr.msg.empty()) fail(herr.msg, herr.line ? herr.line : n.line, herr.line ? herr.col : n.col, herr.status ? herr.status : ST_SCRIPT);
            return out;
        }
        fail("Function not found: " + signature(n.text, args), n);
    }

    Value eval(const Node& n)
    {
        tick(n);
        if (!stack_left()) fail("Stack overflow", n);
        switch (n.k) {
        case NK::IntLit: return Value::from_int(n.ival);
        case NK::FloatLit: return Value::from_float(n.fval);
        case NK::BoolLit: return Value::from_bool(n.ival != 0);
        case NK::StrLit: return Value::from_str(n.text);
        case NK::Interp: {
            std::string s;
            for (const auto& k : n.kids) s += eval(*k).to_string();
            if (s.size() > MAX_STRING) fail("Length of string too large", n);
            return Value::from_str(s);
        }
        case NK::ArrayLit: {
            std::vector<Value> v;
            bool nested = false;
            for (const auto& k : n.kids) {
                v.push_back(eval(*k).copy());
                nested |= v.back().t == Value::Array;
            }
            Value lit = Value::from_array(std::move(v));
            if (nested) check_array(lit, n);
            return lit;
        }
        case NK::Var: {
            Var* v = find(n.text);
            if (!v) fail("Variable not found: " + n.text, n);
            return v->v;
        }
        case NK::Unary: {
            Value v = eval(*n.kids[0]);
            if (n.text == "!") return Value::from_bool(!truthy(v, n));
            if (v.t == Value::Int) {
                if (v.i == std::numeric_limits<int64_t>::min()) fail("Negation overflow: -" + std::to_string(v.i), n);
                return Value::from_int(-v.i);
            }
            if (v.t == Value::Float) return Value::from_float(-v.f);
            fail(std::string("Function not found: - (") + v.type_name() + ")", n);
        }
        case NK::Binary: {
            Value a = eval(*n.kids[0]);
            Value b = eval(*n.kids[1]);
            return binary_op(n.text, a, b, n);
        }
        case NK::And: {
            if (!truthy(eval(*n.kids[0]), n)) return Value::from_bool(false);
            return Value::from_bool(truthy(eval(*n.kids[1]), n));
        }
        case NK::Or: {
            if (truthy(eval(*n.kids[0]), n)) return Value::from_bool(true);
            return Value::from_bool(truthy(eval(*n.kids[1]), n));
        }
        case NK::RangeLit: {
            Value a = eval(*n.kids[0]), b = eval(*n.kids[1]);
            if (a.t != Value::Int || b.t != Value::Int) fail("range bounds must be integers", n);
            Value r;
            r.t = Value::Range;
            r.i = a.i;
            r.j = n.flag ? (b.i == std::numeric_limits<int64_t>::max() ? b.i : b.i + 1) : b.i;
            r.f = 1.0;
            return r;
        }
        case NK::Call: {
            if (n.ival == 1) { // property getter
                std::vector<Value> a{eval(*n.kids[0])};
                Value out;
                if ((n.text == "len" || n.text == "is_empty") && builtin(n.text, a, out, n, nullptr)) return out;
                fail("Unknown property '" + n.text + "' - a getter is not registered for type '" + a[0].type_name() + "'", n);
            }
            return call(n);
        }
        case NK::Index: {
            Value base = eval(*n.kids[0]);
            Value idx = eval(*n.kids[1]);
            if (base.t == Value::Array) {
                if (idx.t != Value::Int) fail(std::string("Data type incorrect: ") + idx.type_name() + " (expecting i64)", n);
                const int64_t len = (int64_t)base.a->size();
                const int64_t i = idx.i < 0 ? len + idx.i : idx.i;
                if (i < 0 || i >= len) fail("Array index " + std::to_string(idx.i) + " out of bounds: only " + std::to_string(len) + " elements in the array", n);
                return (*base.a)[(size_t)i];
            }
            fail(std::string("Indexing is not supported for type ") + base.type_name(), n, base.t == Value::Str ? ST_UNSUPPORTED : ST_SCRIPT);
        }
        case NK::ClosureLit: {
            Value v;
            v.t = Value::Fn;
            v.fn = std::make_shared<Closure>();
            v.fn->params = n.params;
            v.fn->body = n.kids[0];
            std::set<std::string> ids;
            collect_idents(n.kids[0], ids);
            for (const std::string& id : ids) {
                bool is_param = false;
                for (const auto& p : n.params) is_param = is_param || p == id;
                if (is_param) continue;
                if (Var* var = find(id)) v.fn->captured[id] = var->v.copy();
            }
            return v;
        }
        case NK::If: {
            if (truthy(eval(*n.kids[0]), *n.kids[0])) return block(*n.kids[1]);
            if (n.kids.size() > 2) return n.kids[2]->k == NK::If ? eval(*n.kids[2]) : block(*n.kids[2]);
            return Value();
        }
        case NK::Block: return block(n);
        case NK::Let: {
            Value v = n.kids.empty() ? Value() : eval(*n.kids[0]).copy();
            scopes.back()[n.text] = {v, n.ival == 1};
            return Value();
        }
        case NK::Assign: {
            Value rhs = eval(*n.kids[1]);
            bool is_const = false;
            Value* dst = lvalue(*n.kids[0], is_const);
            if (is_const) fail("Cannot modify constant: " + (n.kids[0]->k == NK::Var ? n.kids[0]->text : std::string("<indexed>")), n);
            if (n.text == "=") *dst = rhs.copy();
            else *dst = binary_op(n.text.substr(0, n.text.size() - 1), *dst, rhs, n);
            if (n.kids[0]->k == NK::Index && dst->t == Value::Array) { // an array stored into an element: the whole container is what the limit measures
                const Node* root = n.kids[0].get();
                while (root->k == NK::Index) root = root->kids[0].get();
                if (Var* holder = root->k == NK::Var ? find(root->text) : nullptr) check_array(holder->v, n);
            }
            return Value();
        }
        case NK::While: {
            while (truthy(eval(*n.kids[0]), *n.kids[0])) {
                try { block(*n.kids[1]); } catch (Brk&) { break; } catch (Cont&) { continue; }
            }
            return Value();
        }
        case NK::Loop: {
            for (;;) {
                tick(n);
                try { block(*n.kids[0]); } catch (Brk&) { break; } catch (Cont&) { continue; }
            }
            return Value();
        }
        case NK::For: {
            Value it = eval(*n.kids[0]);
            auto body = [&](const Value& v) -> bool {
                ScopeGuard g(*this);
                scopes.back()[n.text] = {v, false};
                try { block(*n.kids[1]); } catch (Brk&) { return false; } catch (Cont&) {}
                return true;
            };
            if (it.t == Value::Range) {
                const int64_t step = (int64_t)it.f;
                if (step > 0) { for (int64_t v = it.i; v < it.j; v += step) { tick(n); if (!body(Value::from_int(v))) break; if (v > std::numeric_limits<int64_t>::max() - step) break; } }
                else if (step < 0) { for (int64_t v = it.i; v > it.j; v += step) { tick(n); if (!body(Value::from_int(v))) break; if (v < std::numeric_limits<int64_t>::min() - step) break; } }
            } else if (it.t == Value::Array) {
                const std::vector<Value> items = *it.a;
                for (const Value& v : items) { tick(n); if (!body(v)) break; }
            } else fail(std::string("For loop expects an iterable type, not ") + it.type_name(), n);
            return Value();
        }
        case NK::Break: throw Brk{};
        case NK::Continue: throw Cont{};
        case NK::Return: throw Ret{n.kids.empty() ? Value() : eval(*n.kids[0])};
        case NK::FnDef: return Value();
        case NK::ExprStmt: return eval(*n.kids[0]);
        case NK::DoWhile: {
            for (;;) {
                tick(n);
                try { block(*n.kids[0]); } catch (Brk&) { break; } catch (Cont&) {}
                if (truthy(eval(*n.kids[1]), *n.kids[1]) == n.flag) break; // `while`: stop when false; `until`: stop when true
            }
            return Value();
        }
        case NK::Throw: { // EvalAltResult::ErrorRuntime(value, pos): "Runtime error" / "Runtime error: <value>"
            Value v = n.kids.empty() ? Value() : eval(*n.kids[0]).copy();
            const std::string text = v.t == Value::Unit ? std::string() : v.to_string();
            throw rhai::Throw{{text.empty() ? std::string("Runtime error") : "Runtime error: " + text, n.line, n.col, ST_SCRIPT}, std::make_shared<Value>(v)};
        }
        case NK::Try: {
            try {
                return block(*n.kids[0]);
            } catch (rhai::Throw& t) {
                // not catchable, as in Rhai: the sandbox limits and anything outside the supported subset
                static const char* fatal[] = {"Too many operations", "Stack overflow", "Length of string too large", "Size of array too large"};
                for (const char* f : fatal) if (t.e.msg == f) throw;
                if (t.e.status == ST_UNSUPPORTED) throw;
                ScopeGuard g(*this);
                // the thrown value; for a runtime error Rhai binds an object map, here (no maps in the subset) its message
                if (!n.text.empty()) scopes.back()[n.text] = {t.value ? *t.value : Value::from_str(t.e.msg), false};
                return block(*n.kids[1], false);
            }
        }
        case NK::Switch: {
            const Value v = eval(*n.kids[0]);
            for (size_t k = 1; k < n.kids.size(); ++k) {
                const Node& arm = *n.kids[k];
                bool hit = arm.text == "_";
                for (int64_t pi = 0; pi < arm.ival && !hit; ++pi) {
                    const Node& pat = *arm.kids[(size_t)pi];
                    if (pat.k == NK::RangeLit) { // integer ranges match integers only
                        const Value r = eval(pat);
                        hit = v.t == Value::Int && r.t == Value::Range && v.i >= r.i && v.i < r.j;
                    } else {
                        const Value c = eval(pat);
                        hit = c.t == v.t && values_equal(c, v); // cases are matched by type and value: 1 does not match 1.0
                    }
                }
                if (!hit) continue;
                if (arm.flag && !truthy(eval(*arm.kids[(size_t)arm.ival]), *arm.kids[(size_t)arm.ival])) continue;
                const Node& body = *arm.kids.back();
                return body.k == NK::Block ? block(body) : eval(body);
            }
            return Value();
        }
        case NK::Arm: return Value();
        }
        return Value();
    }
};

namespace {
struct RunCall { Interp* in; const char* source; Error* err; bool ok; };
}

// The body of a run, on the calling thread.  Everything a script can raise is a Throw; anything else (allocation failure) becomes an error too,
// because this may be the top frame of a thread.
bool Interp::run_here(const char* source, Error& err)
{
    StackFloorScope floor;
    try {
        if (on_run_thread) on_run_thread();
        Parser p(source);
        NodeP prog = p.program();
        for (const auto& st : prog->kids)
            if (st->k == NK::FnDef) fns_[st->text + "/" + std::to_string(st->params.size())] = st;
        Eval ev(*this);
        try {
            ev.block(*prog, false);
        } catch (Eval::Ret&) { // top-level `return` ends the script
        } catch (Eval::Brk&) {
            fail("'break' outside of a loop", 0, 0);
        } catch (Eval::Cont&) {
            fail("'continue' outside of a loop", 0, 0);
        }
        return true;
    } catch (Throw& t) {
        err = t.e;
        if (!err.status) err.status = ST_SCRIPT;
        return false;
    } catch (const std::exception& e) {
        err = {std::string("script runtime: ") + e.what(), 0, 0, ST_SCRIPT};
        return false;
    }
}

bool Interp::run(const char* source, Error& err)
{
    {   // a caller that already stands on a stack of the size this would create needs no thread (the frames are still checked against that stack's low end)
        const uintptr_t floor_here = this_threads_stack_floor(), sp = (uintptr_t)__builtin_frame_address(0);
        if (floor_here != 0 && sp > floor_here && sp - floor_here >= RUN_STACK_BYTES - (4u << 20)) return run_here(source, err);
    }
    RunCall call{this, source, &err, false};
    pthread_attr_t at;
    pthread_t th;
    bool started = false;
    if (pthread_attr_init(&at) == 0) {
        if (pthread_attr_setstacksize(&at, RUN_STACK_BYTES) == 0)   // reserved address space; pages are touched only as deep as the script goes
            started = pthread_create(&th, &at, [](void* q) -> void* {
                auto* c = static_cast<RunCall*>(q);
                c->ok = c->in->run_here(c->source, *c->err);
                return nullptr;
            }, &call) == 0;
        pthread_attr_destroy(&at);
    }
    if (!started) return run_here(source, err); // no address space for the stack: the caller's stack, still guarded
    pthread_join(th, nullptr);
    return call.ok;
}

bool Interp::call_closure(const Closure& c, std::vector<Value>& args, Value& out, Error& err)
{
    StackFloorScope floor;
    try {
        Eval ev(*this);
        Node at;
        at.k = NK::Block;
        out = ev.call_closure(c, args, at);
        return true;
    } catch (Throw& t) {
        err = t.e;
        return false;
    } catch (Eval::Ret& r) {
        out = r.v;
        return true;
    }
}

// =========================================================================================================== closure -> bytecode
namespace {

enum class CT : uint8_t { I, F, B, U, A };
struct CVal {
    CT t = CT::U;
    int reg = -1;
    std::vector<CVal> elems; // CT::A: compile-time tuple
};

struct Comp {
    Interp& in;
    const std::map<std::string, NodeP>& fns;
    BcProgram& prog;
    int64_t img_w, img_h;
    int next_reg;
    int inline_depth = 0;
    struct VarSlot { CVal v; };
    std::vector<std::map<std::string, VarSlot>> scopes;
    struct LoopCtx { std::vector<size_t> breaks, continues; size_t top = 0; bool is_for = false; };
    std::vector<LoopCtx> loops;
    struct InlineCtx { int result_reg; CT result_t; bool typed; std::vector<size_t> exits; };
    std::vector<InlineCtx> inlines;

    Comp(Interp& i, const std::map<std::string, NodeP>& f, BcProgram& p, int64_t w, int64_t h) : in(i), fns(f), prog(p), img_w(w), img_h(h), next_reg(p.n_params) { scopes.emplace_back(); }

    [[noreturn]] void unsupported(const std::string& what, const Node& n)
    {
        fail("per-pixel closure: " + what + " cannot be compiled for the GPU (supported: i64/f64/bool arithmetic, let, if/else, while/for-range loops, "
             "calls to pure math / pixel-read functions, a 4-element array result)", n, ST_UNSUPPORTED);
    }
    int alloc(const Node& n)
    {
        if (next_reg >= 120) unsupported("an expression this large", n);
        const int r = next_reg++;
        if (next_reg > prog.n_regs) prog.n_regs = next_reg;
        return r;
    }
    size_t emit(uint16_t op, int dst, int a, int b, int c, const Node& n)
    {
        if (prog.code.size() >= 60000) unsupported("a closure this long", n);
        prog.code.push_back({op, (uint16_t)dst, (uint16_t)a, (uint16_t)b, (uint16_t)c, (uint16_t)std::min(n.line, 65535)});
        return prog.code.size() - 1;
    }
    int konst(uint64_t bits)
    {
        for (size_t k = 0; k < prog.consts.size(); ++k) if (prog.consts[k] == bits) return (int)k;
        prog.consts.push_back(bits);
        return (int)prog.consts.size() - 1;
    }
    CVal load_int(int64_t v, const Node& n) { CVal c; c.t = CT::I; c.reg = alloc(n); emit(BC_LOADK, c.reg, konst((uint64_t)v), 0, 0, n); return c; }
    CVal load_float(double v, const Node& n) { uint64_t b; std::memcpy(&b, &v, 8); CVal c; c.t = CT::F; c.reg = alloc(n); emit(BC_LOADK, c.reg, konst(b), 0, 0, n); return c; }
    CVal load_bool(bool v, const Node& n) { CVal c = load_int(v ? 1 : 0, n); c.t = CT::B; return c; }
    CVal to_float(const CVal& v, const Node& n)
    {
        if (v.t == CT::F) return v;
        CVal c; c.t = CT::F; c.reg = alloc(n);
        emit(BC_I2F, c.reg, v.reg, 0, 0, n);
        return c;
    }
    VarSlot* find(const std::string& name)
    {
        for (size_t k = scopes.size(); k-- > 0;) {
            auto it = scopes[k].find(name);
            if (it != scopes[k].end()) return &it->second;
        }
        return nullptr;
    }
    CVal from_value(const Value& v, const Node& n)
    {
        switch (v.t) {
        case Value::Int: return load_int(v.i, n);
        case Value::Float: return load_float(v.f, n);
        case Value::Bool: return load_bool(v.b, n);
        case Value::Unit: { CVal c; c.t = CT::U; return c; }
        case Value::Array: {
            CVal c; c.t = CT::A;
            for (const Value& e : *v.a) {
                if (e.t == Value::Array) unsupported("a nested captured array", n);
                c.elems.push_back(from_value(e, n));
            }
            return c;
        }
        default: unsupported(std::string("a captured value of type ") + v.type_name(), n);
        }
    }
    static const char* tname(CT t) { return t == CT::I ? "i64" : t == CT::F ? "f64" : t == CT::B ? "bool" : t == CT::U ? "()" : "array"; }
    [[noreturn]] void fn_not_found(const std::string& name, const std::vector<CVal>& a, const Node& n)
    {
        std::string s = name + " (";
        for (size_t k = 0; k < a.size(); ++k) { if (k) s += ", "; s += tname(a[k].t); }
        fail("Function not found: " + s + ")", n);
    }

    CVal arith(const std::string& op, CVal a, CVal b, const Node& n)
    {
        const bool cmp = op == "==" || op == "!=" || op == "<" || op == "<=" || op == ">" || op == ">=";
        auto num = [](CT t) { return t == CT::I || t == CT::F; };
        if (a.t == CT::B && b.t == CT::B) {
            CVal r; r.t = CT::B; r.reg = alloc(n);
            if (op == "==") { emit(BC_IEQ, r.reg, a.reg, b.reg, 0, n); return r; }
            if (op == "!=" || op == "^") { emit(BC_INE, r.reg, a.reg, b.reg, 0, n); return r; }
            if (op == "&") { emit(BC_IAND, r.reg, a.reg, b.reg, 0, n); return r; }
            if (op == "|") { emit(BC_IOR, r.reg, a.reg, b.reg, 0, n); return r; }
        }
        if (!num(a.t) || !num(b.t)) {
            if (op == "==" && a.t != b.t) return load_bool(false, n);
            if (op == "!=" && a.t != b.t) return load_bool(true, n);
            fail(std::string("Function not found: ") + op + " (" + tname(a.t) + ", " + tname(b.t) + ")", n);
        }
        CVal r;
        r.reg = alloc(n);
        if (a.t == CT::I && b.t == CT::I) {
            r.t = cmp ? CT::B : CT::I;
            static const std::map<std::string, uint16_t> ops = {{"+", BC_IADD}, {"-", BC_ISUB}, {"*", BC_IMUL}, {"/", BC_IDIV}, {"%", BC_IMOD}, {"**", BC_IPOW},
                                                                {"&", BC_IAND}, {"|", BC_IOR}, {"^", BC_IXOR}, {"<<", BC_ISHL}, {">>", BC_ISHR}, {"==", BC_IEQ},
                                                                {"!=", BC_INE}, {"<", BC_ILT}, {"<=", BC_ILE}, {">", BC_IGT}, {">=", BC_IGE}};
            auto it = ops.find(op);
            if (it == ops.end()) unsupported("operator '" + op + "'", n);
            emit(it->second, r.reg, a.reg, b.reg, 0, n);
            return r;
        }
        a = to_float(a, n);
        b = to_float(b, n);
        r.t = cmp ? CT::B : CT::F;
        static const std::map<std::string, uint16_t> fops = {{"+", BC_FADD}, {"-", BC_FSUB}, {"*", BC_FMUL}, {"/", BC_FDIV}, {"%", BC_FMOD}, {"**", BC_FPOW},
                                                             {"==", BC_FEQ}, {"!=", BC_FNE}, {"<", BC_FLT}, {"<=", BC_FLE}, {">", BC_FGT}, {">=", BC_FGE}};
        auto it = fops.find(op);
        if (it == fops.end()) fail(std::string("Function not found: ") + op + " (f64, f64)", n);
        emit(it->second, r.reg, a.reg, b.reg, 0, n);
        return r;
    }

    CVal call(const Node& n)
    {
        std::vector<CVal> a;
        for (const auto& k : n.kids) a.push_back(expr(*k));
        const std::string& f = n.text;
        const size_t na = a.size();
        auto all = [&](CT t) { for (const CVal& v : a) if (v.t != t) return false; return true; };
        auto op1 = [&](uint16_t op, CT rt) { CVal r; r.t = rt; r.reg = alloc(n); emit(op, r.reg, a[0].reg, 0, 0, n); return r; };
        auto op2 = [&](uint16_t op, CT rt) { CVal r; r.t = rt; r.reg = alloc(n); emit(op, r.reg, a[0].reg, a[1].reg, 0, n); return r; };
        auto op3 = [&](uint16_t op, CT rt) { CVal r; r.t = rt; r.reg = alloc(n); emit(op, r.reg, a[0].reg, a[1].reg, a[2].reg, n); return r; };
        // script-defined function: inline it
        auto fit = fns.find(f + "/" + std::to_string(na));
        if (fit != fns.end()) return inline_fn(*fit->second, a, n);
        if (VarSlot* vs = find(f)) { (void)vs; unsupported("calling a closure stored in a variable", n); }
        if (f == "width" && na == 0) return load_int(img_w, n);
        if (f == "height" && na == 0) return load_int(img_h, n);
        if (f == "PI" && na == 0) return load_float(3.14159265358979323846, n);
        if ((f == "abs" || f == "abs_i") && na == 1 && a[0].t == CT::I) return op1(BC_IABS, CT::I);
        if (f == "abs" && na == 1 && a[0].t == CT::F) return op1(BC_FABS, CT::F);
        if (f == "sign" && na == 1 && a[0].t == CT::I) return op1(BC_ISIGN, CT::I);
        if ((f == "min" || f == "min_i") && na == 2 && all(CT::I)) return op2(BC_IMIN, CT::I);
        if ((f == "max" || f == "max_i") && na == 2 && all(CT::I)) return op2(BC_IMAX, CT::I);
        if ((f == "min" || f == "min_f") && na == 2 && all(CT::F)) return op2(BC_FMIN, CT::F);
        if ((f == "max" || f == "max_f") && na == 2 && all(CT::F)) return op2(BC_FMAX, CT::F);
        if (f == "clamp" && na == 3 && all(CT::I)) return op3(BC_ICLAMP, CT::I);
        if (f == "clamp_f" && na == 3 && all(CT::F)) return op3(BC_FCLAMP, CT::F);
        if (f == "lerp" && na == 3 && all(CT::F)) return op3(BC_FLERP, CT::F);
        if (f == "distance" && na == 4 && all(CT::F)) {
            CVal r; r.t = CT::F; r.reg = alloc(n);
            // operands must be consecutive for the 4-operand form: copy
            const int base = alloc(n); alloc(n); alloc(n); alloc(n);
            for (int k = 0; k < 4; ++k) emit(BC_MOV, base + k, a[k].reg, 0, 0, n);
            emit(BC_FDIST, r.reg, base, 0, 0, n);
            return r;
        }
        if (na == 1 && a[0].t == CT::F) {
            static const std::map<std::string, uint16_t> f1 = {{"floor", BC_FFLOOR}, {"ceil", BC_FCEIL}, {"round", BC_FROUND}, {"sqrt", BC_FSQRT}, {"sin", BC_FSIN},
                                                               {"cos", BC_FCOS}, {"tan", BC_FTAN}, {"exp", BC_FEXP}, {"ln", BC_FLN}};
            auto it = f1.find(f);
            if (it != f1.end()) return op1(it->second, CT::F);
        }
        if (f == "pow" && na == 2 && all(CT::F)) return op2(BC_FPOW, CT::F);
        if (f == "atan2" && na == 2 && all(CT::F)) return op2(BC_FATAN2, CT::F);
        if (f == "to_float" && na == 1 && (a[0].t == CT::I || a[0].t == CT::F)) return to_float(a[0], n);
        if (f == "to_int" && na == 1 && a[0].t == CT::I) return a[0];
        if (f == "to_int" && na == 1 && a[0].t == CT::F) return op1(BC_F2I, CT::I);
        if (f == "is_selected" && na == 2 && all(CT::I)) return op2(BC_ISSEL, CT::B);
        if ((f == "get_r" || f == "get_g" || f == "get_b" || f == "get_a") && na == 2 && all(CT::I)) {
            CVal r; r.t = CT::I; r.reg = alloc(n);
            emit(BC_GETCH, r.reg, a[0].reg, a[1].reg, f == "get_r" ? 0 : f == "get_g" ? 1 : f == "get_b" ? 2 : 3, n);
            return r;
        }
        if (f == "get_pixel" && na == 2 && all(CT::I)) {
            CVal r; r.t = CT::A;
            for (int c = 0; c < 4; ++c) { CVal e; e.t = CT::I; e.reg = alloc(n); emit(BC_GETCH, e.reg, a[0].reg, a[1].reg, c, n); r.elems.push_back(e); }
            return r;
        }
        if (f == "len" && na == 1 && a[0].t == CT::A) return load_int((int64_t)a[0].elems.size(), n);
        for (const char* impure : {"set_pixel", "set_r", "set_g", "set_b", "set_a", "rand_int", "rand_float", "print", "print_line", "debug", "sleep", "progress", "for_each_pixel",
                                   "map_channels", "for_region", "push", "pop", "select_rect", "select_ellipse", "fill_selected", "delete_selected", "clear_selection",
                                   "invert_selection", "has_selection", "rgb_to_hsl", "hsl_to_rgb"})
            if (f == impure) unsupported("a call to '" + f + "' (state-changing or array-valued host function)", n);
        if (f.rfind("apply_", 0) == 0 || f.rfind("flip_", 0) == 0 || f.rfind("rotate_", 0) == 0 || f.rfind("resize_", 0) == 0)
            unsupported("a call to '" + f + "'", n);
        fn_not_found(f, a, n);
    }

    CVal inline_fn(const Node& def, const std::vector<CVal>& args, const Node& at)
    {
        if (++inline_depth > 12) unsupported("recursive or deeply nested script functions", at);
        std::vector<std::map<std::string, VarSlot>> saved;
        saved.swap(scopes);
        scopes.emplace_back();
        for (size_t k = 0; k < def.params.size(); ++k) { // parameters are copies
            CVal p = args[k];
            if (p.t != CT::A && p.t != CT::U) { CVal c; c.t = p.t; c.reg = alloc(at); emit(BC_MOV, c.reg, p.reg, 0, 0, at); p = c; }
            scopes.back()[def.params[k]] = {p};
        }
        InlineCtx ic;
        ic.result_reg = alloc(at);
        ic.result_t = CT::U;
        ic.typed = false;
        inlines.push_back(ic);
        std::vector<LoopCtx> saved_loops;
        saved_loops.swap(loops);
        CVal v = block(*def.kids[0], false);
        InlineCtx& cur = inlines.back();
        store_result(cur, v, at);
        for (size_t j : cur.exits) prog.code[j].a = (uint16_t)prog.code.size();
        CVal out;
        out.t = cur.result_t;
        out.reg = cur.result_reg;
        inlines.pop_back();
        loops.swap(saved_loops);
        scopes.swap(saved);
        --inline_depth;
        return out;
    }
    void store_result(InlineCtx& ic, const CVal& v, const Node& at)
    {
        if (v.t == CT::A) unsupported("a script function returning an array", at);
        if (ic.typed && ic.result_t != v.t) unsupported("a script function whose return type depends on the path taken", at);
        ic.typed = true;
        ic.result_t = v.t;
        if (v.t != CT::U) emit(BC_MOV, ic.result_reg, v.reg, 0, 0, at);
    }

    CVal expr(const Node& n)
    {
        switch (n.k) {
        case NK::IntLit: return load_int(n.ival, n);
        case NK::FloatLit: return load_float(n.fval, n);
        case NK::BoolLit: return load_bool(n.ival != 0, n);
        case NK::ArrayLit: {
            CVal r; r.t = CT::A;
            for (const auto& k : n.kids) r.elems.push_back(expr(*k));
            return r;
        }
        case NK::Var: {
            VarSlot* v = find(n.text);
            if (!v) fail("Variable not found: " + n.text, n);
            return v->v;
        }
        case NK::Unary: {
            CVal v = expr(*n.kids[0]);
            CVal r; r.reg = alloc(n);
            if (n.text == "!") {
                if (v.t != CT::B) fail(std::string("Data type incorrect: ") + tname(v.t) + " (expecting bool)", n);
                r.t = CT::B; emit(BC_NOT, r.reg, v.reg, 0, 0, n); return r;
            }
            if (v.t == CT::I) { r.t = CT::I; emit(BC_INEG, r.reg, v.reg, 0, 0, n); return r; }
            if (v.t == CT::F) { r.t = CT::F; emit(BC_FNEG, r.reg, v.reg, 0, 0, n); return r; }
            fail(std::string("Function not found: - (") + tname(v.t) + ")", n);
        }
        case NK::Binary: {
            CVal a = expr(*n.kids[0]);
            CVal b = expr(*n.kids[1]);
            return arith(n.text, a, b, n);
        }
        case NK::And:
        case NK::Or: {
            CVal r; r.t = CT::B; r.reg = alloc(n);
            CVal a = expr(*n.kids[0]);
            if (a.t != CT::B) fail(std::string("Data type incorrect: ") + tname(a.t) + " (expecting bool)", n);
            emit(BC_MOV, r.reg, a.reg, 0, 0, n);
            const size_t j = emit(n.k == NK::And ? BC_JZ : BC_JNZ, 0, 0, r.reg, 0, n); // a = target (patched), b = condition register
            CVal b = expr(*n.kids[1]);
            if (b.t != CT::B) fail(std::string("Data type incorrect: ") + tname(b.t) + " (expecting bool)", n);
            emit(BC_MOV, r.reg, b.reg, 0, 0, n);
            prog.code[j].a = (uint16_t)prog.code.size();
            return r;
        }
        case NK::Call:
            if (n.ival == 1) {
                CVal v = expr(*n.kids[0]);
                if (n.text == "len" && v.t == CT::A) return load_int((int64_t)v.elems.size(), n);
                unsupported("property '" + n.text + "'", n);
            }
            return call(n);
        case NK::Index: {
            CVal base = expr(*n.kids[0]);
            if (base.t != CT::A) fail(std::string("Indexing is not supported for type ") + tname(base.t), n);
            const Node& ix = *n.kids[1];
            if (ix.k != NK::IntLit) unsupported("an array index that is not an integer literal", n);
            const int64_t len = (int64_t)base.elems.size();
            const int64_t i = ix.ival < 0 ? len + ix.ival : ix.ival;
            if (i < 0 || i >= len) fail("Array index " + std::to_string(ix.ival) + " out of bounds: only " + std::to_string(len) + " elements in the array", n);
            return base.elems[(size_t)i];
        }
        case NK::If: {
            CVal c = expr(*n.kids[0]);
            if (c.t != CT::B) fail(std::string("Data type incorrect: ") + tname(c.t) + " (expecting bool)", *n.kids[0]);
            const int res = alloc(n);
            const size_t jz = emit(BC_JZ, 0, 0, c.reg, 0, n);
            CVal t = block(*n.kids[1], true);
            if (t.t == CT::A) unsupported("an array-valued if-expression that is not the closure's result", n);
            if (t.t != CT::U) emit(BC_MOV, res, t.reg, 0, 0, n);
            const size_t jend = emit(BC_JMP, 0, 0, 0, 0, n);
            prog.code[jz].a = (uint16_t)prog.code.size();
            CVal e;
            if (n.kids.size() > 2) {
                e = n.kids[2]->k == NK::If ? expr(*n.kids[2]) : block(*n.kids[2], true);
                if (e.t == CT::A) unsupported("an array-valued if-expression that is not the closure's result", n);
                if (e.t != CT::U) emit(BC_MOV, res, e.reg, 0, 0, n);
            }
            prog.code[jend].a = (uint16_t)prog.code.size();
            CVal r;
            if (t.t == e.t) { r.t = t.t; r.reg = res; }
            else r.t = CT::U; // used as a statement; a value use of mismatched branches is a type error at the use site
            return r;
        }
        case NK::Block: return block(n, true);
        case NK::StrLit:
        case NK::Interp: unsupported("a string", n);
        case NK::ClosureLit: unsupported("a nested closure", n);
        case NK::RangeLit: unsupported("a range value outside a for loop", n);
        default: unsupported("this statement in expression position", n);
        }
    }

    void assign(const Node& n)
    {
        if (n.kids[0]->k != NK::Var) unsupported("assignment to an array element", n);
        VarSlot* v = find(n.kids[0]->text);
        if (!v) fail("Variable not found: " + n.kids[0]->text, n);
        CVal rhs = expr(*n.kids[1]);
        if (n.text != "=") rhs = arith(n.text.substr(0, n.text.size() - 1), v->v, rhs, n);
        if (v->v.t == CT::A || rhs.t == CT::A) {
            if (rhs.t != CT::A || v->v.t != CT::A) unsupported("changing a variable between array and scalar", n);
            if (!loops.empty() || rhs.elems.size() != v->v.elems.size()) unsupported("re-assigning an array variable inside a loop", n);
            // copy element-wise so that later reads see the new registers' values
            for (size_t k = 0; k < rhs.elems.size(); ++k) {
                if (rhs.elems[k].t != v->v.elems[k].t) unsupported("changing an array element's type", n);
                emit(BC_MOV, v->v.elems[k].reg, rhs.elems[k].reg, 0, 0, n);
            }
            return;
        }
        if (rhs.t != v->v.t) unsupported(std::string("changing the type of variable '") + n.kids[0]->text + "' from " + tname(v->v.t) + " to " + tname(rhs.t), n);
        if (rhs.t != CT::U) emit(BC_MOV, v->v.reg, rhs.reg, 0, 0, n);
    }

    // statements; returns the value of a trailing expression
    CVal block(const Node& b, bool new_scope)
    {
        if (new_scope) scopes.emplace_back();
        CVal last;
        for (size_t k = 0; k < b.kids.size(); ++k) {
            const Node& st = *b.kids[k];
            const bool is_last = k + 1 == b.kids.size();
            const int mark = next_reg;
            CVal v = stmt(st);
            if (is_last && st.k == NK::ExprStmt && !st.flag) last = v;
            else {
                last = CVal();
                // temporaries die with the statement; `let` keeps what it allocated
                const bool keeps = st.k == NK::ExprStmt && st.kids[0]->k == NK::Let;
                if (!keeps) next_reg = mark;
            }
        }
        if (new_scope) scopes.pop_back();
        return last;
    }

    CVal stmt(const Node& st)
    {
        if (st.k == NK::FnDef) return CVal();
        const Node& n = st.k == NK::ExprStmt ? *st.kids[0] : st;
        switch (n.k) {
        case NK::Let: {
            CVal v;
            if (!n.kids.empty()) v = expr(*n.kids[0]);
            if (v.t == CT::A) { // own copies so that later assignments cannot alias the initialiser
                CVal c; c.t = CT::A;
                for (const CVal& e : v.elems) {
                    if (e.t == CT::A || e.t == CT::U) unsupported("a nested array", n);
                    CVal x; x.t = e.t; x.reg = alloc(n); emit(BC_MOV, x.reg, e.reg, 0, 0, n); c.elems.push_back(x);
                }
                v = c;
            } else if (v.t != CT::U) {
                CVal c; c.t = v.t; c.reg = alloc(n); emit(BC_MOV, c.reg, v.reg, 0, 0, n); v = c;
            }
            scopes.back()[n.text] = {v};
            return CVal();
        }
        case NK::Assign: assign(n); return CVal();
        case NK::While:
        case NK::Loop: {
            LoopCtx lc;
            lc.top = prog.code.size();
            loops.push_back(lc);
            size_t jz = (size_t)-1;
            if (n.k == NK::While) {
                CVal c = expr(*n.kids[0]);
                if (c.t != CT::B) fail(std::string("Data type incorrect: ") + tname(c.t) + " (expecting bool)", *n.kids[0]);
                jz = emit(BC_JZ, 0, 0, c.reg, 0, n);
            }
            block(*n.kids[n.k == NK::While ? 1 : 0], true);
            emit(BC_JMP, 0, (int)loops.back().top, 0, 0, n);
            const size_t end = prog.code.size();
            if (jz != (size_t)-1) prog.code[jz].a = (uint16_t)end;
            for (size_t j : loops.back().breaks) prog.code[j].a = (uint16_t)end;
            loops.pop_back();
            return CVal();
        }
        case NK::For: {
            // for v in a..b | a..=b | range(a, b[, step])
            const Node& it = *n.kids[0];
            CVal lo, hi, step;
            bool inclusive = false, have_step = false;
            if (it.k == NK::RangeLit) { lo = expr(*it.kids[0]); hi = expr(*it.kids[1]); inclusive = it.flag; }
            else if (it.k == NK::Call && it.text == "range" && (it.kids.size() == 2 || it.kids.size() == 3)) {
                lo = expr(*it.kids[0]); hi = expr(*it.kids[1]);
                if (it.kids.size() == 3) { step = expr(*it.kids[2]); have_step = true; }
            } else unsupported("a for loop over anything but an integer range", n);
            if (lo.t != CT::I || hi.t != CT::I || (have_step && step.t != CT::I)) fail("range bounds must be integers", n);
            if (have_step && it.kids[2]->k != NK::IntLit) unsupported("a range step that is not an integer literal", n);
            const int64_t stepv = have_step ? it.kids[2]->ival : 1;
            if (stepv == 0) fail("range: step is zero", n);
            scopes.emplace_back();
            CVal var; var.t = CT::I; var.reg = alloc(n);
            emit(BC_MOV, var.reg, lo.reg, 0, 0, n);
            CVal end; end.t = CT::I; end.reg = alloc(n);
            emit(BC_MOV, end.reg, hi.reg, 0, 0, n);
            CVal stepc = load_int(stepv, n);
            scopes.back()[n.text] = {var};
            LoopCtx lc;
            const size_t top = prog.code.size();
            CVal c; c.t = CT::B; c.reg = alloc(n);
            emit(stepv > 0 ? (inclusive ? BC_ILE : BC_ILT) : BC_IGT, c.reg, var.reg, end.reg, 0, n);
            const size_t jz = emit(BC_JZ, 0, 0, c.reg, 0, n);
            // the loop variable is a fresh copy per iteration in Rhai: body assignments to it do not affect iteration
            CVal shadow; shadow.t = CT::I; shadow.reg = alloc(n);
            emit(BC_MOV, shadow.reg, var.reg, 0, 0, n);
            scopes.back()[n.text] = {shadow};
            lc.is_for = true; // `continue` jumps to the increment (patched below)
            loops.push_back(lc);
            block(*n.kids[1], true);
            const size_t incr = prog.code.size();
            emit(BC_IADD, var.reg, var.reg, stepc.reg, 0, n); // an overflowing increment raises "Addition overflow" only at i64::MAX bounds
            emit(BC_JMP, 0, (int)top, 0, 0, n);
            const size_t endpc = prog.code.size();
            prog.code[jz].a = (uint16_t)endpc;
            for (size_t j : loops.back().breaks) prog.code[j].a = (uint16_t)endpc;
            for (size_t j : loops.back().continues) prog.code[j].a = (uint16_t)incr;
            loops.pop_back();
            scopes.pop_back();
            return CVal();
        }
        case NK::Break: {
            if (loops.empty()) fail("'break' outside of a loop", n);
            loops.back().breaks.push_back(emit(BC_JMP, 0, 0, 0, 0, n));
            return CVal();
        }
        case NK::Continue: {
            if (loops.empty()) fail("'continue' outside of a loop", n);
            if (loops.back().is_for) loops.back().continues.push_back(emit(BC_JMP, 0, 0, 0, 0, n));
            else emit(BC_JMP, 0, (int)loops.back().top, 0, 0, n);
            return CVal();
        }
        case NK::Return: {
            CVal v;
            if (!n.kids.empty()) v = expr(*n.kids[0]);
            if (!inlines.empty()) {
                store_result(inlines.back(), v, n);
                inlines.back().exits.push_back(emit(BC_JMP, 0, 0, 0, 0, n));
            } else ret(v, n);
            return CVal();
        }
        default: return expr(n);
        }
    }

    void ret(const CVal& v, const Node& n)
    {
        if (v.t == CT::A && v.elems.size() >= 4) {
            const int base = alloc(n); alloc(n); alloc(n); alloc(n);
            int mask = 0;
            for (int k = 0; k < 4; ++k) {
                if (v.elems[k].t == CT::I) { mask |= 1 << k; emit(BC_MOV, base + k, v.elems[k].reg, 0, 0, n); }
                else if (v.elems[k].t == CT::A) unsupported("a nested array", n);
            }
            emit(BC_RET_ARR, 0, base, mask, 0, n);
        } else emit(BC_RET_UNIT, 0, 0, 0, 0, n);
    }

    // tail position: the closure's result
    void tail(const Node& n)
    {
        if (n.k == NK::Block) {
            scopes.emplace_back();
            for (size_t k = 0; k < n.kids.size(); ++k) {
                const Node& st = *n.kids[k];
                const bool is_last = k + 1 == n.kids.size();
                if (is_last && st.k == NK::ExprStmt && !st.flag) { tail(*st.kids[0]); scopes.pop_back(); return; }
                const int mark = next_reg;
                stmt(st);
                if (!(st.k == NK::ExprStmt && st.kids[0]->k == NK::Let)) next_reg = mark;
            }
            scopes.pop_back();
            emit(BC_RET_UNIT, 0, 0, 0, 0, n);
            return;
        }
        if (n.k == NK::If) {
            CVal c = expr(*n.kids[0]);
            if (c.t != CT::B) fail(std::string("Data type incorrect: ") + tname(c.t) + " (expecting bool)", *n.kids[0]);
            const size_t jz = emit(BC_JZ, 0, 0, c.reg, 0, n);
            tail(*n.kids[1]);
            prog.code[jz].a = (uint16_t)prog.code.size();
            if (n.kids.size() > 2) tail(*n.kids[2]);
            else emit(BC_RET_UNIT, 0, 0, 0, 0, n);
            return;
        }
        if (n.k == NK::Return) { CVal v; if (!n.kids.empty()) v = expr(*n.kids[0]); ret(v, n); return; }
        if (n.k == NK::While || n.k == NK::Loop || n.k == NK::For || n.k == NK::Let || n.k == NK::Assign) { stmt(n); emit(BC_RET_UNIT, 0, 0, 0, 0, n); return; }
        ret(expr(n), n);
    }
};
} // namespace

// Constants out of the per-pixel program.  The interpreter's cost is per instruction (k_script.hip: ~80 scalar instructions and ~30 branches of dispatch each),
// and a typical closure — `[255 - r, g / 2, (b * 3 + a) / 4, a]` — spends 3 of its 10 instructions loading literals.  In a STRAIGHT-LINE program (no jumps: a
// register's contents at any instruction are decided by the instructions in front of it) every LOADK moves into a preamble that fills a register of its own once
// per lane — the register file persists over the lane's pixels — and the reads of the LOADK's destination, up to the next write of it, read that register
// instead.  Operands that must be consecutive registers (FDIST, the result array) get the value moved back first.  Programs with jumps are left as they are.
namespace {
void hoist_constants(BcProgram& p)
{
    for (const BcIns& I : p.code) if (I.op == BC_JMP || I.op == BC_JZ || I.op == BC_JNZ) return;
    std::vector<BcIns> pre, body;
    std::map<uint16_t, int> reg_of_const;
    std::vector<int> alias(256, -1);                       // alias[r] = the hoisted register that holds what r would hold
    int n_regs = p.n_regs;
    auto rd = [&](uint16_t r) -> uint16_t { return (r < alias.size() && alias[r] >= 0) ? (uint16_t)alias[r] : r; };
    auto writes = [&](uint16_t r) { if (r < alias.size()) alias[r] = -1; };
    auto materialize = [&](uint16_t r, uint16_t line) {
        if (r < alias.size() && alias[r] >= 0) { body.push_back({BC_MOV, r, (uint16_t)alias[r], 0, 0, line}); alias[r] = -1; }
    };
    for (BcIns I : p.code) {
        switch (I.op) {
        case BC_LOADK: {
            auto it = reg_of_const.find(I.a);
            if (it == reg_of_const.end()) {
                if (n_regs >= 120) { writes(I.dst); body.push_back(I); break; }   // the register file is full: this one stays a per-pixel load
                it = reg_of_const.emplace(I.a, n_regs++).first;
                pre.push_back({BC_LOADK, (uint16_t)it->second, I.a, 0, 0, I.line});
            }
            if (I.dst < alias.size()) alias[I.dst] = it->second;
            break;
        }
        case BC_FDIST:                                       // reads a .. a + 3
            for (int k = 0; k < 4; ++k) materialize((uint16_t)(I.a + k), I.line);
            writes(I.dst);
            body.push_back(I);
            break;
        case BC_RET_ARR:                                     // reads a .. a + 3; b is a bit mask
            for (int k = 0; k < 4; ++k) materialize((uint16_t)(I.a + k), I.line);
            body.push_back(I);
            break;
        case BC_RET_UNIT: case BC_ERR:                       // no register operands (ERR: a is the error code)
            body.push_back(I);
            break;
        case BC_ICLAMP: case BC_FCLAMP: case BC_FLERP:       // a, b, c
            I.a = rd(I.a); I.b = rd(I.b); I.c = rd(I.c);
            writes(I.dst);
            body.push_back(I);
            break;
        case BC_GETCH:                                       // a, b; c is the channel
        default:                                             // unary (a) and binary (a, b) operations, MOV, GETCH, ISSEL: an unused field is rewritten harmlessly only
            I.a = rd(I.a);                                   // if it names a register — unary ops carry b = 0, which is parameter register 0 and never aliased
            I.b = rd(I.b);
            writes(I.dst);
            body.push_back(I);
            break;
        }
    }
    if (pre.empty()) return;
    p.n_pre = (int)pre.size();
    p.n_regs = n_regs;
    pre.insert(pre.end(), body.begin(), body.end());
    p.code.swap(pre);
}
} // namespace

bool Interp::compile_closure(const Closure& c, int n_params, int64_t img_w, int64_t img_h, BcProgram& out, Error& err)
{
    try {
        Node at;
        at.k = NK::Block;
        const Closure* cl = &c;
        Closure tmp;
        if (!c.fn_name.empty()) { // Fn("name"): wrap the script function
            auto it = fns_.find(c.fn_name + "/" + std::to_string(n_params));
            if (it == fns_.end()) fail("Function not found: " + c.fn_name + " (" + std::to_string(n_params) + " parameters)", 0, 0);
            tmp.params = it->second->params;
            tmp.body = it->second->kids[0];
            cl = &tmp;
        }
        if ((int)cl->params.size() != n_params) {
            std::string sig = "anon (";
            for (int k = 0; k < n_params; ++k) sig += k ? ", i64" : "i64";
            fail("Function not found: " + sig + ")", cl->body ? cl->body->line : 0, cl->body ? cl->body->col : 0);
        }
        out = BcProgram();
        out.n_params = n_params;
        out.n_regs = n_params;
        Comp comp(*this, fns_, out, img_w, img_h);
        for (int k = 0; k < n_params; ++k) { CVal p; p.t = CT::I; p.reg = k; comp.scopes.back()[cl->params[k]] = {p}; }
        comp.scopes.emplace_back();
        for (const auto& kv : cl->captured) {
            bool is_param = false;
            for (const auto& p : cl->params) is_param = is_param || p == kv.first;
            if (is_param) continue;
            if (kv.second.t == Value::Fn || kv.second.t == Value::Str || kv.second.t == Value::Range) continue; // only an error if actually used
            comp.scopes.back()[kv.first] = {comp.from_value(kv.second, *cl->body)};
        }
        comp.tail(*cl->body);
        hoist_constants(out);
        return true;
    } catch (Throw& t) {
        err = t.e;
        if (!err.status) err.status = ST_SCRIPT;
        return false;
    }
}

const char* Interp::bc_error_text(int code)
{
    switch (code) {
    case BCE_ADD_OVERFLOW: return "Addition overflow";
    case BCE_SUB_OVERFLOW: return "Subtraction overflow";
    case BCE_MUL_OVERFLOW: return "Multiplication overflow";
    case BCE_DIV_ZERO: return "Division by zero";
    case BCE_DIV_OVERFLOW: return "Division overflow";
    case BCE_MOD_ZERO: return "Modulo division by zero";
    case BCE_NEG_OVERFLOW: return "Negation overflow";
    case BCE_POW_OVERFLOW: return "Exponential overflow";
    case BCE_POW_NEGATIVE: return "Integer raised to a negative power";
    case BCE_F2I_RANGE: return "Integer overflow: to_int";
    case BCE_TOO_MANY_OPS: return "Too many operations";
    case BCE_ABS_OVERFLOW: return "Integer overflow: abs";
    default: return "runtime error in per-pixel closure";
    }
}

} // namespace rhai
