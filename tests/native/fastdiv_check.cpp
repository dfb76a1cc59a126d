// Host check of icaf::FastDiv (icafusion_amd/csrc/icaf_common.h): q = (mulhi(n, m) + n) >> s equals n / d for every divisor the
// launchers can build and n over the whole 31-bit range the kernels use it on (strided sweep, every multiple-of-d edge, top of the range).
#include "icaf_common.h"
#include <cstdlib>

int main() {
    using namespace icaf;
    const unsigned int ds[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 14, 16, 18, 20, 40, 42, 80, 100, 160, 255, 256, 320, 400, 640, 1000,
                               1280, 4095, 4096, 65535, 65537, 1u << 20, (1u << 30) + 7, 0x7fffffffu};
    for (unsigned int d : ds) {
        const FastDiv f = make_fastdiv(d);
        for (unsigned long long n = 0; n < (1ull << 31); n += 977)
            if (fd_div((unsigned int)n, f) != (unsigned int)n / d) return std::printf("FAIL d=%u n=%llu\n", d, n), 1;
        for (unsigned int n = 0x7fffffffu - 100000u; n < 0x7fffffffu; ++n)
            if (fd_div(n + 1, f) != (n + 1) / d) return std::printf("FAIL d=%u n=%u\n", d, n + 1), 1;
        for (unsigned int k = 1; k < 200000 && (unsigned long long)k * d < (1ull << 31); ++k) {
            unsigned int q, r;
            fd_divmod(k * d, f, q, r);
            if (q != k || r != 0) return std::printf("FAIL multiple d=%u k=%u\n", d, k), 1;
            fd_divmod(k * d - 1, f, q, r);
            if (q != k - 1 || r != d - 1) return std::printf("FAIL below multiple d=%u k=%u\n", d, k), 1;
        }
    }
    std::srand(1);
    for (int i = 0; i < 2000; ++i) {
        const unsigned int d = (unsigned int)(std::rand() % 100000) + 1;
        const FastDiv f = make_fastdiv(d);
        for (int j = 0; j < 20000; ++j) {
            const unsigned int n = ((unsigned int)std::rand() * 2654435761u) & 0x7fffffffu;
            if (fd_div(n, f) != n / d) return std::printf("FAIL random d=%u n=%u\n", d, n), 1;
        }
    }
    std::puts("fastdiv ok");
    return 0;
}
