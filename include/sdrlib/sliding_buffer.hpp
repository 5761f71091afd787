// sliding_buffer.hpp -- drop-in for the reference's utils::sliding_buffer
// (lib/inc/sliding_buffer.hpp:14-104): a fixed-size window over a stream; what does not fit
// wraps around, index 0 is the oldest element still held.
//
// Same names and behaviour: explicit sliding_buffer(size) zero-fills (:52-54); operator[](pos)
// reads (cursor + pos) mod size (:73-76); at(pos) is the checked form returning
// expected<T, std::string> with the text "pos=<pos> exceeds size=<size>" (:59-65);
// push_back(first, last) / push_back(value) write at the cursor and advance it with wrap-around
// (:78-94); begin()/end() walk positions 0..size (:56-57); size() (:96-99).
// Deliberate difference: a range LONGER than the buffer is accepted here (only its tail
// survives, as if pushed element by element); the reference's single wrap (:84-86) would run
// past the end of its storage in that case.
// Host-side only: this is the window the demo keeps for its plots and running text
// (app/QFDemoWindow.cpp:20-21); the device-side counterpart is sdrlib/pipe.hpp.
#pragma once

#include <cstddef>
#include <iterator>
#include <string>
#include <vector>

#include "expected.hpp"

namespace utils {

template <typename T>
class sliding_buffer {
public:
    class iterator {
    public:
        using value_type = T;
        using difference_type = std::ptrdiff_t;
        using pointer = T*;
        using reference = T&;
        using iterator_category = std::forward_iterator_tag;

        iterator() = default;
        iterator(sliding_buffer* owner, std::size_t pos) : owner_(owner), pos_(pos) {}
        reference operator*() const { return (*owner_)[pos_]; }
        pointer operator->() const { return &(*owner_)[pos_]; }
        iterator& operator++() { ++pos_; return *this; }
        iterator operator++(int) { iterator before = *this; ++pos_; return before; }
        bool operator==(const iterator& o) const { return pos_ == o.pos_; }
        bool operator!=(const iterator& o) const { return pos_ != o.pos_; }

    private:
        sliding_buffer* owner_ = nullptr;
        std::size_t pos_ = 0;
    };

    explicit sliding_buffer(std::size_t size) : store_(size, T{}), cursor_(0) {}

    iterator begin() { return iterator(this, 0); }
    iterator end() { return iterator(this, store_.size()); }

    sdrlib::expected<T, std::string> at(std::size_t pos) const
    {
        if (pos >= store_.size())
            return sdrlib::unexpected<std::string>("pos=" + std::to_string(pos) + " exceeds size=" + std::to_string(store_.size()));
        return (*this)[pos];
    }

    const T& operator[](std::size_t pos) const { return store_[(cursor_ + pos) % store_.size()]; }
    T& operator[](std::size_t pos) { return store_[(cursor_ + pos) % store_.size()]; }

    template <typename It>
    void push_back(It first, It last)
    {
        if (store_.empty()) return;
        for (; first != last; ++first) {
            store_[cursor_] = *first;
            cursor_ = (cursor_ + 1 == store_.size()) ? 0 : cursor_ + 1;
        }
    }

    void push_back(const T& value) { push_back(&value, &value + 1); }

    std::size_t size() const { return store_.size(); }

private:
    std::vector<T> store_;
    std::size_t cursor_;
};

} // namespace utils
