// host_test.cpp -- host-only pieces of the drop-in headers (no device needed):
// utils::sliding_buffer against the behaviour of lib/inc/sliding_buffer.hpp:14-104 as the demo
// uses it (app/QFDemoWindow.cpp:20-21, 222, 229-235, 264).
#include <cstdio>
#include <numeric>
#include <string>
#include <vector>

#include "sdrlib/sliding_buffer.hpp"

static int g_failed = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failed; } } while (0)

// The reference's own nine cases (lib/tests/sliding_buffer_test.cpp:11-154), restated: same
// inputs, same expected windows.
static void reference_cases()
{
    using utils::sliding_buffer;
    { sliding_buffer<int> b(5); CHECK(b.size() == 5u); }                                   // ConstructWithSize (:11-15)
    {                                                                                       // IndexOutOfBounds (:21-28)
        sliding_buffer<int> b(3);
        auto r = b.at(3);
        CHECK(!r.has_value());
        CHECK(r.error().find("exceeds size") != std::string::npos);
    }
    {                                                                                       // PushSingleElement (:34-43)
        sliding_buffer<int> b(4);
        b.push_back(42);
        auto v = b.at(3);
        CHECK(v.has_value() && *v == 42);
    }
    {                                                                                       // PushRangeNoWrap (:49-60)
        sliding_buffer<int> b(5);
        std::vector<int> src{1, 2, 3};
        b.push_back(src.begin(), src.end());
        CHECK(b[2] == 1 && b[3] == 2 && b[4] == 3);
    }
    {                                                                                       // PushRangeExactFit (:66-77)
        sliding_buffer<int> b(4);
        std::vector<int> src{10, 20, 30, 40};
        b.push_back(src.begin(), src.end());
        CHECK(b[0] == 10 && b[1] == 20 && b[2] == 30 && b[3] == 40);
    }
    {                                                                                       // PushRangeWithWrap (:83-100)
        sliding_buffer<int> b(5);
        std::vector<int> first{1, 2, 3, 4}, second{5, 6, 7};
        b.push_back(first.begin(), first.end());
        b.push_back(second.begin(), second.end());
        CHECK(b[0] == 3 && b[1] == 4 && b[2] == 5 && b[3] == 6 && b[4] == 7);
    }
    {                                                                                       // OverwriteOldestData (:106-116)
        sliding_buffer<int> b(3);
        std::vector<int> src{1, 2, 3, 4, 5};
        b.push_back(src.begin(), src.end());
        CHECK(b[0] == 3 && b[1] == 4 && b[2] == 5);
    }
    {                                                                                       // MultipleSmallPushes (:122-136)
        sliding_buffer<int> b(4);
        for (int v = 1; v <= 5; ++v) b.push_back(v);
        CHECK(b[0] == 2 && b[1] == 3 && b[2] == 4 && b[3] == 5);
    }
    {                                                                                       // IndexIsRelativeToCurrentHead (:142-154)
        sliding_buffer<int> b(3);
        for (int v : {10, 20, 30, 40}) b.push_back(v);
        CHECK(b[0] == 20 && b[1] == 30 && b[2] == 40);
    }
}

int main()
{
    reference_cases();
    utils::sliding_buffer<int> b(5);
    CHECK(b.size() == 5);
    for (std::size_t i = 0; i < 5; ++i) CHECK(b[i] == 0);                  // zero filled (:52-54)

    std::vector<int> a{1, 2, 3};
    b.push_back(a.begin(), a.end());                                       // cursor -> 3
    // index 0 is the cursor: the two untouched zeros come first, then what was pushed
    CHECK(b[0] == 0 && b[1] == 0 && b[2] == 1 && b[3] == 2 && b[4] == 3);

    std::vector<int> c{4, 5, 6, 7};
    b.push_back(c.begin(), c.end());                                       // wraps: storage 6 7 3 4 5, cursor 2
    CHECK(b[0] == 3 && b[1] == 4 && b[2] == 5 && b[3] == 6 && b[4] == 7);
    b.push_back(8);
    CHECK(b[0] == 4 && b[4] == 8);

    // exact fill up to the end of storage wraps the cursor to 0 (:85-87)
    utils::sliding_buffer<int> e(4);
    std::vector<int> four{1, 2, 3, 4};
    e.push_back(four.begin(), four.end());
    CHECK(e[0] == 1 && e[3] == 4);

    // checked access (:59-65)
    auto ok = b.at(4);
    CHECK(ok.has_value() && *ok == 8);
    auto bad = b.at(5);
    CHECK(!bad.has_value() && bad.error() == "pos=5 exceeds size=5");

    // iteration visits positions 0..size in window order (:56-57); the demo builds its text so (:264)
    utils::sliding_buffer<unsigned char> t(6);
    const std::string s = "abcdefgh";
    t.push_back(s.begin(), s.end());                                       // longer than the window: tail survives
    CHECK(std::string(t.begin(), t.end()) == "cdefgh");
    *t.begin() = 'X';
    CHECK(t[0] == 'X');
    int n = 0;
    for (auto it = t.begin(); it != t.end(); it++) ++n;
    CHECK(n == 6);

    utils::sliding_buffer<int> z(0);                                       // degenerate size: nothing to write to
    z.push_back(1);
    CHECK(z.size() == 0 && !z.at(0).has_value());

    if (g_failed) { std::printf("%d host checks FAILED\n", g_failed); return 1; }
    std::printf("all host checks passed\n");
    return 0;
}
