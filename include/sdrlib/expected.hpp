// expected.hpp -- result type of the drop-in headers.
//
// The reference returns std::expected<T, std::string> (lib/inc/fft.hpp:88, ofdm.hpp:14,38).
// Where the standard library provides <expected> (C++23) the alias below IS std::expected;
// otherwise a small own implementation with the members the reference's callers use
// (has_value, operator bool, operator*, operator->, value, error, value_or, and_then,
// transform, or_else) stands in.  It lives in namespace sdrlib so that it can never clash with
// libstdc++'s legacy std::unexpected() function.
#pragma once

#include <exception>
#include <functional>
#include <string>
#include <type_traits>
#include <utility>
#include <version>

#if defined(__cpp_lib_expected) && __cpp_lib_expected >= 202202L
#include <expected>
namespace sdrlib {
template <class T, class E> using expected = std::expected<T, E>;
template <class E> using unexpected = std::unexpected<E>;
}
#else

namespace sdrlib {

template <class E>
class unexpected {
public:
    constexpr explicit unexpected(E e) : err_(std::move(e)) {}
    template <class G, class = std::enable_if_t<std::is_constructible_v<E, G&&> && !std::is_same_v<std::decay_t<G>, unexpected>>>
    constexpr explicit unexpected(G&& g) : err_(std::forward<G>(g)) {}
    constexpr const E& error() const& noexcept { return err_; }
    constexpr E& error() & noexcept { return err_; }
    constexpr E&& error() && noexcept { return std::move(err_); }
private:
    E err_;
};
template <class E> unexpected(E) -> unexpected<E>;
unexpected(const char*) -> unexpected<std::string>;

class bad_expected_access : public std::exception {
public:
    const char* what() const noexcept override { return "bad access to sdrlib::expected without a value"; }
};

template <class T, class E> class expected;

namespace detail {
template <class X> struct is_expected : std::false_type {};
template <class T, class E> struct is_expected<expected<T, E>> : std::true_type {};
}

template <class T, class E>
class expected {
public:
    using value_type = T;
    using error_type = E;

    expected() : has_(true) { new (&val_) T(); }
    expected(const T& v) : has_(true) { new (&val_) T(v); }
    expected(T&& v) : has_(true) { new (&val_) T(std::move(v)); }
    template <class G> expected(const unexpected<G>& u) : has_(false) { new (&err_) E(u.error()); }
    template <class G> expected(unexpected<G>&& u) : has_(false) { new (&err_) E(std::move(u).error()); }
    expected(const expected& o) : has_(o.has_) { if (has_) new (&val_) T(o.val_); else new (&err_) E(o.err_); }
    expected(expected&& o) noexcept : has_(o.has_) { if (has_) new (&val_) T(std::move(o.val_)); else new (&err_) E(std::move(o.err_)); }
    expected& operator=(const expected& o) { if (this != &o) { destroy(); has_ = o.has_; if (has_) new (&val_) T(o.val_); else new (&err_) E(o.err_); } return *this; }
    expected& operator=(expected&& o) noexcept { if (this != &o) { destroy(); has_ = o.has_; if (has_) new (&val_) T(std::move(o.val_)); else new (&err_) E(std::move(o.err_)); } return *this; }
    ~expected() { destroy(); }

    bool has_value() const noexcept { return has_; }
    explicit operator bool() const noexcept { return has_; }
    T& operator*() & { return val_; }
    const T& operator*() const& { return val_; }
    T&& operator*() && { return std::move(val_); }
    T* operator->() { return &val_; }
    const T* operator->() const { return &val_; }
    T& value() & { if (!has_) throw bad_expected_access(); return val_; }
    const T& value() const& { if (!has_) throw bad_expected_access(); return val_; }
    T&& value() && { if (!has_) throw bad_expected_access(); return std::move(val_); }
    E& error() & { return err_; }
    const E& error() const& { return err_; }
    template <class U> T value_or(U&& d) const& { return has_ ? val_ : static_cast<T>(std::forward<U>(d)); }

    template <class F> auto and_then(F&& f) & { using R = std::remove_cv_t<std::remove_reference_t<std::invoke_result_t<F, T&>>>; static_assert(detail::is_expected<R>::value); return has_ ? std::invoke(std::forward<F>(f), val_) : R(unexpected<E>(err_)); }
    template <class F> auto and_then(F&& f) && { using R = std::remove_cv_t<std::remove_reference_t<std::invoke_result_t<F, T&&>>>; static_assert(detail::is_expected<R>::value); return has_ ? std::invoke(std::forward<F>(f), std::move(val_)) : R(unexpected<E>(std::move(err_))); }
    template <class F> auto transform(F&& f) & {
        using U = std::remove_cv_t<std::invoke_result_t<F, T&>>;
        if constexpr (std::is_void_v<U>) { if (has_) { std::invoke(std::forward<F>(f), val_); return expected<void, E>(); } return expected<void, E>(unexpected<E>(err_)); }
        else { return has_ ? expected<U, E>(std::invoke(std::forward<F>(f), val_)) : expected<U, E>(unexpected<E>(err_)); }
    }
    template <class F> auto transform(F&& f) && {
        using U = std::remove_cv_t<std::invoke_result_t<F, T&&>>;
        if constexpr (std::is_void_v<U>) { if (has_) { std::invoke(std::forward<F>(f), std::move(val_)); return expected<void, E>(); } return expected<void, E>(unexpected<E>(std::move(err_))); }
        else { return has_ ? expected<U, E>(std::invoke(std::forward<F>(f), std::move(val_))) : expected<U, E>(unexpected<E>(std::move(err_))); }
    }
    template <class F> expected or_else(F&& f) const& { return has_ ? *this : std::invoke(std::forward<F>(f), err_); }

private:
    void destroy() { if (has_) val_.~T(); else err_.~E(); }
    bool has_;
    union { T val_; E err_; };
};

template <class E>
class expected<void, E> {
public:
    using value_type = void;
    using error_type = E;

    expected() : has_(true) {}
    template <class G> expected(const unexpected<G>& u) : has_(false), err_(u.error()) {}
    template <class G> expected(unexpected<G>&& u) : has_(false), err_(std::move(u).error()) {}

    bool has_value() const noexcept { return has_; }
    explicit operator bool() const noexcept { return has_; }
    void operator*() const noexcept {}
    void value() const { if (!has_) throw bad_expected_access(); }
    E& error() & { return err_; }
    const E& error() const& { return err_; }

    template <class F> auto and_then(F&& f) const { using R = std::remove_cv_t<std::remove_reference_t<std::invoke_result_t<F>>>; static_assert(detail::is_expected<R>::value); return has_ ? std::invoke(std::forward<F>(f)) : R(unexpected<E>(err_)); }
    template <class F> auto transform(F&& f) const {
        using U = std::remove_cv_t<std::invoke_result_t<F>>;
        if constexpr (std::is_void_v<U>) { if (has_) { std::invoke(std::forward<F>(f)); return expected<void, E>(); } return expected<void, E>(unexpected<E>(err_)); }
        else { return has_ ? expected<U, E>(std::invoke(std::forward<F>(f))) : expected<U, E>(unexpected<E>(err_)); }
    }
    template <class F> expected or_else(F&& f) const { return has_ ? *this : std::invoke(std::forward<F>(f), err_); }

private:
    bool has_;
    E err_{};
};

} // namespace sdrlib
#endif
