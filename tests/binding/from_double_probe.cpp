// Reads "I F S Q O n v1 ... vn" lines (hex floats) and prints the raw values QuBLAS_amd.h's double constructor
// logic (detail::from_double) produces, one line per input line.
#include "QuBLAS_amd.h"

#include <cstdio>
#include <cstdlib>

int main()
{
    int I, F, S, Q, O, n;
    while (std::scanf("%d %d %d %d %d %d", &I, &F, &S, &Q, &O, &n) == 6) {
        QuBLAS_amd::Fmt f{I, F, S != 0, Q, O};
        for (int i = 0; i < n; ++i) {
            char buf[64];
            if (std::scanf("%63s", buf) != 1) return 2;
            std::printf("%s%lld", i ? " " : "", (long long)QuBLAS_amd::detail::from_double(std::strtod(buf, nullptr), f));
        }
        std::printf("\n");
    }
    return 0;
}
