// ascii_entry() (the table entries computed from immediates) == the tables built from the letter lists,
// for every byte of every table.  Plain C++ (tests/test_oracle_ascii.py compiles and runs it with g++).
#include <cstdint>
#include <cstdio>

#include "../../kmers.jl_amd/csrc/ascii_tables.hpp"

int main() {
    using namespace kmers;
    uint8_t t[256];
    int bad = 0;
    for (int table = 0; table < 4; ++table) {
        build_ascii_encode_table(table >= 2 ? 4 : 2, (table & 1) != 0, t);
        for (int c = 0; c < 256; ++c)
            if (ascii_entry((uint32_t)table, (uint32_t)c) != t[c]) {
                std::printf("table %d byte 0x%02x: %d vs %d\n", table, c, ascii_entry((uint32_t)table, (uint32_t)c), t[c]);
                ++bad;
            }
    }
    build_ascii_skipping_table(t);
    for (int c = 0; c < 256; ++c)
        if (ascii_entry((uint32_t)ASCII_TABLE_SKIPPING, (uint32_t)c) != t[c]) {
            std::printf("skipping table byte 0x%02x: %d vs %d\n", c, ascii_entry((uint32_t)ASCII_TABLE_SKIPPING, (uint32_t)c), t[c]);
            ++bad;
        }
    std::printf(bad ? "MISMATCHES %d\n" : "ascii_entry agrees with the tables (%d)\n", bad);
    return bad != 0;
}
