// Tahoe/Math/Error.h -- assertion / logging macros of the facade.
// Reference: Tahoe/Math/Error.h:24-58 (ADLASSERT is gtest EXPECT_TRUE under TH_UNIT_TEST, a no-op in a
// plain release build, a trap in _DEBUG).  Here a failed ADLASSERT is never silent: under
// TH_UNIT_TEST with gtest it is EXPECT_TRUE, otherwise it reports on stderr and bumps
// adl_assert_failures() so that a harness without gtest can still fail its run.
#pragma once
#include <stdio.h>

inline int& adl_assert_failures()
{
    static int count = 0;
    return count;
}

#if defined(TH_UNIT_TEST) && defined(GTEST_INCLUDE_GTEST_GTEST_H_)
#define ADLASSERT(x) EXPECT_TRUE(x)
#else
#define ADLASSERT(x)                                                                  \
    do {                                                                              \
        if (!(x)) {                                                                   \
            if (adl_assert_failures()++ < 20)                                         \
                fprintf(stderr, "ADLASSERT failed: %s (%s:%d)\n", #x, __FILE__, __LINE__); \
        }                                                                             \
    } while (0)
#endif
#define ADLWARN(x) { x; }
#define ADLCOMPILEASSERT(x) static_assert(x, "CompileTimeAssert")

#ifndef TH_LOG_ERROR
#define TH_LOG_ERROR(...) fprintf(stderr, __VA_ARGS__)
#endif
#ifndef TH_LOG_DEBUG
#define TH_LOG_DEBUG(...) ((void)0)
#endif
#define debugPrintf(...) ((void)0)
