// (re)defines assert() for the reference's test sources: throw instead of abort, accept mask arrays.  Included again by
// the shim headers that may pull in <cassert> after test.h did.
#undef assert
#define assert(...)                                                                                                  \
    do {                                                                                                             \
        if (!test::truth(__VA_ARGS__))                                                                               \
            throw test::Failure(std::string("assertion failed at line ") + std::to_string(__LINE__) + ": " #__VA_ARGS__); \
    } while (0)
