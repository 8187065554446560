// jrt.hpp — a minimal Java-runtime stand-in for the MECHANICALLY TRANSLATED reference classes (test infrastructure).
//
// tools/make_ref.py rewrites the reference's Java decision classes token by token into C++ (oracle/_ref/gen/); this header
// supplies what those tokens expect from the JDK: object references with null checks (Ref<T> -> NullPointerException),
// arrays (JArr<T>), the exception taxonomy (Exception vs Error matters: `catch (Exception e)` does NOT catch an
// AssertionError), atomics, the few collections, Math / Long / Integer / System / String.format and a scope guard for
// `finally`.  Nothing here knows anything about Raft.  Single-threaded by design: the driver serialises every event the
// way one EventLoop thread would.  Compile with -fwrapv (Java integer arithmetic wraps).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <initializer_list>
#include <map>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

typedef int64_t jlong;
typedef int32_t jint;
typedef int16_t jshort;
typedef int8_t  jbyte;
typedef bool    jboolean;
typedef double  jdouble;

// ---- objects and references ---------------------------------------------------------------------------------------
struct Class;
struct Object {
    int rc_ = 1;                         // born pinned: jnew drops the pin once the constructor has returned
    Object() {}
    Object(const Object &) : rc_(1) {}
    Object &operator=(const Object &) { return *this; }
    virtual ~Object() {}
    virtual Class *klass_() { return nullptr; }      // java.lang.Object.getClass() for the translated classes
};

struct String;
template <class T> struct Ref;
typedef Ref<String> JString;

struct Throwable : virtual Object {      // thrown by value (and, as a callback argument, passed by reference)
    std::string msg;
    const char *where = "";              // "File.java:line" of the `throw` in the reference (set by jat)
    Throwable() {}
    explicit Throwable(const std::string &m) : msg(m) {}
    virtual ~Throwable() {}
    virtual const char *kind() const { return "Throwable"; }
};
template <class E> E jat(E e, const char *where) { e.where = where; return e; }
#define JRT_THROWABLE(Name, Base)                                                        \
    struct Name : Base {                                                                 \
        Name() {}                                                                        \
        Name(const std::string &m) : Base(m) {}                                          \
        Name(const char *m) : Base(std::string(m)) {}                                    \
        explicit Name(const Throwable &cause) : Base(cause.msg) {}                        \
        template <class S> Name(const Ref<S> &s);                                        \
        const char *kind() const override { return #Name; }                              \
    };
JRT_THROWABLE(Exception, Throwable)
JRT_THROWABLE(RuntimeException, Exception)
JRT_THROWABLE(NullPointerException, RuntimeException)
JRT_THROWABLE(IllegalStateException, RuntimeException)
JRT_THROWABLE(IllegalArgumentException, RuntimeException)
JRT_THROWABLE(IndexOutOfBoundsException, RuntimeException)
JRT_THROWABLE(ArrayIndexOutOfBoundsException, IndexOutOfBoundsException)
JRT_THROWABLE(ClassCastException, RuntimeException)
JRT_THROWABLE(Error, Throwable)
JRT_THROWABLE(AssertionError, Error)
JRT_THROWABLE(LinkageError, Error)
JRT_THROWABLE(IncompatibleClassChangeError, LinkageError)
JRT_THROWABLE(AbstractMethodError, IncompatibleClassChangeError)

template <class T> struct Ref {
    T *p = nullptr;
    Object *o = nullptr;                 // the same object seen as Object: lets copies/destruction work on incomplete T
    Ref() {}
    Ref(std::nullptr_t) {}
    Ref(T *q) : p(q), o(q) { if (o) o->rc_++; }
    Ref(const Ref &r) : p(r.p), o(r.o) { if (o) o->rc_++; }
    Ref(Ref &&r) noexcept : p(r.p), o(r.o) { r.p = nullptr; r.o = nullptr; }
    template <class U, class = typename std::enable_if<std::is_convertible<U *, T *>::value>::type>
    Ref(const Ref<U> &r) : p(r.p), o(r.o) { if (o) o->rc_++; }
    ~Ref() { drop(); }
    Ref &operator=(const Ref &r) { if (r.o) r.o->rc_++; drop(); p = r.p; o = r.o; return *this; }
    Ref &operator=(Ref &&r) noexcept { if (this != &r) { drop(); p = r.p; o = r.o; r.p = nullptr; r.o = nullptr; } return *this; }
    void drop() { if (o && --o->rc_ == 0) delete o; p = nullptr; o = nullptr; }
    T *operator->() const { if (!p) throw NullPointerException("null dereference"); return p; }
    T &operator*() const { if (!p) throw NullPointerException("null dereference"); return *p; }
    T *get() const { return p; }
    explicit operator bool() const { return p != nullptr; }
};
template <class A, class B> bool operator==(const Ref<A> &a, const Ref<B> &b) { return (const void *)a.o == (const void *)b.o; }
template <class A, class B> bool operator!=(const Ref<A> &a, const Ref<B> &b) { return (const void *)a.o != (const void *)b.o; }
template <class A> bool operator==(const Ref<A> &a, std::nullptr_t) { return a.p == nullptr; }
template <class A> bool operator!=(const Ref<A> &a, std::nullptr_t) { return a.p != nullptr; }
template <class A> bool operator==(std::nullptr_t, const Ref<A> &a) { return a.p == nullptr; }
template <class A> bool operator!=(std::nullptr_t, const Ref<A> &a) { return a.p != nullptr; }
template <class A, class B> bool operator==(const Ref<A> &a, const B *b) { return (const void *)a.p == (const void *)b; }
template <class A, class B> bool operator!=(const Ref<A> &a, const B *b) { return (const void *)a.p != (const void *)b; }
template <class A, class B> bool operator==(const B *b, const Ref<A> &a) { return (const void *)a.p == (const void *)b; }
template <class A, class B> bool operator!=(const B *b, const Ref<A> &a) { return (const void *)a.p != (const void *)b; }

// `new T(args)`: the object is pinned while its constructor runs (a constructor may hand `this` out)
template <class T, class... A> Ref<T> jnew(A &&... a)
{
    T *t = new T(std::forward<A>(a)...);
    Ref<T> r(t);
    static_cast<Object *>(t)->rc_--;
    return r;
}
template <class T, class U> Ref<T> jcast(const Ref<U> &r)
{
    if (!r.p) return Ref<T>();
    T *t = dynamic_cast<T *>(r.p);
    if (!t) throw ClassCastException("bad cast");
    return Ref<T>(t);
}
template <class T, class U> Ref<T> jcast(U *r) { return jcast<T>(Ref<U>(r)); }

// ---- String ---------------------------------------------------------------------------------------------------------
struct String : virtual Object {
    std::string s;
    String() {}
    String(const std::string &x) : s(x) {}
};
template <> struct Ref<String> {         // value-like: literals convert implicitly
    std::string s; bool null = true;
    Ref() {}
    Ref(std::nullptr_t) {}
    Ref(const char *c) : s(c), null(false) {}
    Ref(const std::string &c) : s(c), null(false) {}
};
#define JRT_THROWABLE_CTOR(Name) template <class S> Name::Name(const Ref<S> &s) : Name(s.s) {}
JRT_THROWABLE_CTOR(Exception) JRT_THROWABLE_CTOR(RuntimeException) JRT_THROWABLE_CTOR(NullPointerException)
JRT_THROWABLE_CTOR(IllegalStateException) JRT_THROWABLE_CTOR(IllegalArgumentException) JRT_THROWABLE_CTOR(IndexOutOfBoundsException)
JRT_THROWABLE_CTOR(ArrayIndexOutOfBoundsException) JRT_THROWABLE_CTOR(ClassCastException) JRT_THROWABLE_CTOR(Error)
JRT_THROWABLE_CTOR(AssertionError) JRT_THROWABLE_CTOR(LinkageError) JRT_THROWABLE_CTOR(IncompatibleClassChangeError)
JRT_THROWABLE_CTOR(AbstractMethodError)

namespace jrt {
inline void fmt_arg(std::string &out, jlong v) { out += std::to_string(v); }
inline void fmt_arg(std::string &out, jint v) { out += std::to_string(v); }
inline void fmt_arg(std::string &out, bool v) { out += v ? "true" : "false"; }
inline void fmt_arg(std::string &out, const JString &v) { out += v.null ? "null" : v.s; }
template <class T> inline void fmt_arg(std::string &out, const Ref<T> &v) { out += v.p ? "<obj>" : "null"; }
inline void fmt_go(std::string &out, const char *f) { out += f; }
template <class A, class... R> void fmt_go(std::string &out, const char *f, const A &a, const R &... r)
{
    while (*f) {
        if (*f == '%' && f[1] && f[1] != '%') { fmt_arg(out, a); fmt_go(out, f + 2, r...); return; }
        out += *f++;
    }
}
}  // namespace jrt
struct StringStatics {
    template <class... A> static JString format(const char *f, const A &... a) { std::string o; jrt::fmt_go(o, f, a...); return JString(o); }
};
// `String.format(...)` is rewritten to `String_::format(...)`
typedef StringStatics String_;

// ---- arrays ---------------------------------------------------------------------------------------------------------
template <class T> struct JArrBody : Object { jint length = 0; std::vector<T> v; };
template <class T> struct JArr {
    Ref<JArrBody<T>> b;
    JArr() {}
    JArr(std::nullptr_t) {}
    JArr(std::initializer_list<T> l) : b(jnew<JArrBody<T>>()) { b->v.assign(l.begin(), l.end()); b->length = (jint)b->v.size(); }
    static JArr make(jlong n)
    {
        if (n < 0) throw RuntimeException("NegativeArraySizeException");
        JArr a; a.b = jnew<JArrBody<T>>(); a.b->v.assign((size_t)n, T()); a.b->length = (jint)n; return a;
    }
    JArrBody<T> *operator->() const { return b.operator->(); }
    T &operator[](jlong i) const
    {
        JArrBody<T> *q = b.operator->();
        if (i < 0 || i >= q->length) throw ArrayIndexOutOfBoundsException(std::to_string(i));
        return q->v[(size_t)i];
    }
    typename std::vector<T>::iterator begin() const { return b.operator->()->v.begin(); }
    typename std::vector<T>::iterator end() const { return b.operator->()->v.end(); }
};
template <class T> bool operator==(const JArr<T> &a, std::nullptr_t) { return a.b == nullptr; }
template <class T> bool operator!=(const JArr<T> &a, std::nullptr_t) { return a.b != nullptr; }
template <class T> bool operator==(const JArr<T> &a, const JArr<T> &c) { return a.b == c.b; }
template <class T> bool operator!=(const JArr<T> &a, const JArr<T> &c) { return a.b != c.b; }

template <class T> struct is_jarr : std::false_type {};
template <class T> struct is_jarr<JArr<T>> : std::true_type {};
// how a generic container stores an element of (bare) type T: arrays by handle, classes by reference
template <class T> struct slot_of { typedef Ref<T> type; };
template <class T> struct slot_of<JArr<T>> { typedef JArr<T> type; };

struct Arrays {
    template <class T> static JArr<T> copyOfRange(const JArr<T> &a, jint from, jint to)
    {
        if (from < 0 || from > a->length) throw ArrayIndexOutOfBoundsException("copyOfRange");
        if (from > to) throw IllegalArgumentException("from > to");
        JArr<T> r = JArr<T>::make(to - from);
        for (jint i = from; i < to && i < a->length; i++) r[i - from] = a[i];
        return r;
    }
    static void sort(const JArr<jlong> &a) { std::sort(a->v.begin(), a->v.end()); }
    template <class T> static auto asList(const JArr<Ref<T>> &a);
};

// ---- numbers --------------------------------------------------------------------------------------------------------
struct Math {
    static constexpr double E = 2.718281828459045;
    template <class A, class B> static typename std::common_type<A, B>::type max(A a, B b) { typedef typename std::common_type<A, B>::type C; return (C)a > (C)b ? (C)a : (C)b; }
    template <class A, class B> static typename std::common_type<A, B>::type min(A a, B b) { typedef typename std::common_type<A, B>::type C; return (C)a < (C)b ? (C)a : (C)b; }
    static double log(double x) { return std::log(x); }
    // Math.round(double) of Java 7+ (the reference targets 1.8, pom.xml:47-48): floor(x + 1/2) in EXACT arithmetic — not the double
    // addition of Java 6, which rounds 0.49999999999999994 and the odd integers of [2^52, 2^53) up — saturating, NaN -> 0. Restated from
    // the specification ("the long closest to the argument, ties rounding to positive infinity") on the bits of the double.
    static jlong round(double x)
    {
        if (x != x) return 0;
        uint64_t bits; memcpy(&bits, &x, 8);
        const int64_t biased = (int64_t)((bits >> 52) & 0x7FF);
        const int64_t shift = (52 - 1 + 1023) - biased;            // 2^-shift = half an ulp of the significand read as an integer
        if ((shift & ~(int64_t)63) == 0) {                           // 0 <= shift < 64: |x| in [2^-12, 2^52): the fraction can matter
            int64_t r = (int64_t)((bits & 0xFFFFFFFFFFFFFull) | 0x10000000000000ull);
            if ((int64_t)bits < 0) r = -r;
            return ((r >> shift) + 1) >> 1;                          // floor(x * 2) -> floor((x + 1/2) * 2) -> floor(x + 1/2)
        }
        if (x >= 9.2233720368547758e18) return INT64_MAX;            // (long) x of Java: saturating, and exact for |x| >= 2^52; 0 below 2^-12
        if (x <= -9.2233720368547758e18) return INT64_MIN;
        return (jlong)x;
    }
};
struct Long {
    static constexpr jlong MAX_VALUE = INT64_MAX;
    static constexpr jlong MIN_VALUE = INT64_MIN;
    static constexpr jint BYTES = 8;
    static jint compare(jlong a, jlong b) { return a < b ? -1 : (a == b ? 0 : 1); }
    static jint hashCode(jlong v) { return (jint)(v ^ (jlong)((uint64_t)v >> 32)); }
};
struct Integer {
    static constexpr jint MAX_VALUE = INT32_MAX;
    static constexpr jint BYTES = 4;
    static jint compareUnsigned(jint a, jint b) { uint32_t x = (uint32_t)a, y = (uint32_t)b; return x < y ? -1 : (x == y ? 0 : 1); }
};
inline jlong jushr(jlong v, int n) { return (jlong)((uint64_t)v >> (n & 63)); }
inline jint jushr(jint v, int n) { return (jint)((uint32_t)v >> (n & 31)); }

struct Objects {
    template <class T> static T requireNonNull(const T &t) { if (t == nullptr) throw NullPointerException("requireNonNull"); return t; }
};

// ---- the clock (virtual: the driver owns time) ---------------------------------------------------------------------
namespace jrt { inline jlong &now_ms() { static jlong t = 0; return t; } }
struct System { static jlong currentTimeMillis() { return jrt::now_ms(); } };

// ---- logging: every call is a no-op that still evaluates nothing --------------------------------------------------
namespace jrt { inline std::function<void(const char *)> &on_logger_error() { static std::function<void(const char *)> f; return f; } }
struct Logger : virtual Object {
    template <class... A> void debug(const A &...) {}
    template <class... A> void info(const A &...) {}
    template <class... A> void warn(const A &...) {}
    template <class... A> void error(const char *fmt, const A &...) { if (jrt::on_logger_error()) jrt::on_logger_error()(fmt); }   // a swallowed exception
};

// ---- atomics (single-threaded) ---------------------------------------------------------------------------------------
struct AtomicLong : virtual Object {
    jlong v = 0;
    AtomicLong() {}
    AtomicLong(jlong x) : v(x) {}
    virtual jlong get() { return v; }
    virtual void set(jlong x) { v = x; }
    virtual jboolean compareAndSet(jlong e, jlong u) { if (v != e) return false; v = u; return true; }
};
namespace jrt { inline std::function<void(struct ::Object *)> &on_new_atomic_integer() { static std::function<void(Object *)> f; return f; } }
struct AtomicInteger : virtual Object {
    jint v = 0;
    AtomicInteger() { if (jrt::on_new_atomic_integer()) jrt::on_new_atomic_integer()(this); }
    AtomicInteger(jint x) : v(x) { if (jrt::on_new_atomic_integer()) jrt::on_new_atomic_integer()(this); }
    jint get() { return v; }
    void set(jint x) { v = x; }
    jint incrementAndGet() { return ++v; }
    jint decrementAndGet() { return --v; }
};
template <class T> struct AtomicReference : virtual Object {
    Ref<T> v;
    AtomicReference() {}
    Ref<T> get() { return v; }
    void set(const Ref<T> &x) { v = x; }
    jboolean compareAndSet(const Ref<T> &e, const Ref<T> &u) { if (v != e) return false; v = u; return true; }
};
template <class T> struct AtomicLongFieldUpdater : virtual Object {
    jlong T::*f;
    AtomicLongFieldUpdater(jlong T::*m) : f(m) {}
    jboolean compareAndSet(const Ref<T> &o, jlong e, jlong u) { if (o.operator->()->*f != e) return false; o.p->*f = u; return true; }
};
template <class T> struct AtomicIntegerFieldUpdater : virtual Object {
    jint T::*f;
    AtomicIntegerFieldUpdater(jint T::*m) : f(m) {}
    jint incrementAndGet(const Ref<T> &o) { return ++(o.operator->()->*f); }
    jint decrementAndGet(const Ref<T> &o) { return --(o.operator->()->*f); }
};
// `AtomicLongFieldUpdater.newUpdater(X.class, "field")` is rewritten to AtomicLongFieldUpdater_newUpdater(&X::field)
template <class T> Ref<AtomicLongFieldUpdater<T>> AtomicLongFieldUpdater_newUpdater(jlong T::*m) { return jnew<AtomicLongFieldUpdater<T>>(m); }
template <class T> Ref<AtomicIntegerFieldUpdater<T>> AtomicIntegerFieldUpdater_newUpdater(jint T::*m) { return jnew<AtomicIntegerFieldUpdater<T>>(m); }

// ---- collections (only what the translated code touches) -----------------------------------------------------------
template <class F> struct LongStream : virtual Object {
    std::vector<jlong> v;
    JArr<jlong> toArray() { JArr<jlong> a = JArr<jlong>::make((jlong)v.size()); for (size_t i = 0; i < v.size(); i++) a[(jlong)i] = v[i]; return a; }
};
template <class T> struct Stream : virtual Object {
    std::vector<typename slot_of<T>::type> v;
    Ref<LongStream<void>> mapToLong(const std::function<jlong(typename slot_of<T>::type)> &f)
    {
        Ref<LongStream<void>> s = jnew<LongStream<void>>();
        for (auto &e : v) s->v.push_back(f(e));
        return s;
    }
};
template <class T> struct Collection : virtual Object {
    typedef typename slot_of<T>::type E;
    std::vector<E> v;
    virtual jint size() { return (jint)v.size(); }
    virtual jboolean add(const E &e) { v.push_back(e); return true; }
    virtual E get(jint i) { if (i < 0 || i >= (jint)v.size()) throw IndexOutOfBoundsException(std::to_string(i)); return v[(size_t)i]; }
    virtual Ref<Stream<T>> stream() { Ref<Stream<T>> s = jnew<Stream<T>>(); s->v = v; return s; }
    virtual JArr<E> toArray(const JArr<E> &) { JArr<E> a = JArr<E>::make((jlong)v.size()); for (size_t i = 0; i < v.size(); i++) a[(jlong)i] = v[i]; return a; }
    typename std::vector<E>::iterator begin() { return v.begin(); }
    typename std::vector<E>::iterator end() { return v.end(); }
};
template <class T> struct List : Collection<T> {};
template <class T> struct Set : Collection<T> {};
template <class T> struct ArrayList : List<T> { ArrayList() {} ArrayList(jint) {} };
template <class T> auto Arrays::asList(const JArr<Ref<T>> &a) { Ref<List<T>> l = jnew<ArrayList<T>>(); for (auto &e : a) l->add(e); return l; }
template <class T> auto begin(const Ref<T> &r) -> decltype(r->begin()) { return r->begin(); }
template <class T> auto end(const Ref<T> &r) -> decltype(r->end()) { return r->end(); }

// keys compare with K::equals (Java HashMap semantics without the hashing)
template <class K, class V> struct Map : virtual Object {
    std::vector<std::pair<Ref<K>, Ref<V>>> kv;
    virtual Ref<V> get(const Ref<K> &k) { for (auto &e : kv) if (e.first == k || (e.first != nullptr && e.first->equals(k))) return e.second; return nullptr; }
    virtual Ref<V> put(const Ref<K> &k, const Ref<V> &v)
    {
        for (auto &e : kv) if (e.first == k || (e.first != nullptr && e.first->equals(k))) { Ref<V> old = e.second; e.second = v; return old; }
        kv.emplace_back(k, v); return nullptr;
    }
    virtual Ref<V> remove(const Ref<K> &k)
    {
        for (size_t i = 0; i < kv.size(); i++)
            if (kv[i].first == k || (kv[i].first != nullptr && kv[i].first->equals(k))) { Ref<V> old = kv[i].second; kv.erase(kv.begin() + (long)i); return old; }
        return nullptr;
    }
    virtual void clear() { kv.clear(); }
    virtual jint size() { return (jint)kv.size(); }
    virtual Ref<Collection<V>> values() { Ref<Collection<V>> c = jnew<Collection<V>>(); for (auto &e : kv) c->v.push_back(e.second); return c; }
};
template <class K, class V> struct HashMap : Map<K, V> { HashMap() {} HashMap(jint) {} };
template <class K, class V> struct ConcurrentHashMap : Map<K, V> { ConcurrentHashMap() {} };

// ---- try / finally ---------------------------------------------------------------------------------------------------
template <class F> struct JFinally { F f; ~JFinally() noexcept(false) { f(); } };
template <class F> JFinally<F> jfinally(F f) { return JFinally<F>{std::move(f)}; }
