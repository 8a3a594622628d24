// host_lookup.cpp -- the `sorted` step of the lookup argument (kimchi/src/circuits/lookup/constraints.rs:90-194) on the host.
//
// The reference counts, with a HashMap over field elements, how often every entry of the combined table is looked up, repeats each entry
// (count + 1) times in table order and lays the result out as max_per_row + 1 "snake" columns (consecutive columns share one element, every
// second column is reversed).  That is a hash join between two host-sized vectors -- it stays host code here too, but native: the Python loop
// that used to do it cost 0.36 s of a 0.38 s proof at 2^16 rows.  The looked-up values themselves are computed on the device (one expression
// per lookup slot over the resident witness columns and pattern selectors, proof_systems_amd/lookup.py::lookup_values_dev) and come down as
// limbs; nothing here does field arithmetic: values are compared as 32-byte strings (canonical Montgomery limbs are unique).
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../include/kimchi_hip.h"

namespace kh { void set_error(const char* fmt, ...); }

namespace {
struct Key { uint64_t l[4]; };
inline bool same(const Key& a, const Key& b) { return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0; }
inline uint64_t hash(const Key& k) {
    uint64_t h = k.l[0] * 0x9e3779b97f4a7c15ULL ^ k.l[1];
    h = (h ^ (h >> 29)) * 0xbf58476d1ce4e5b9ULL ^ k.l[2];
    h = (h ^ (h >> 32)) * 0x94d049bb133111ebULL ^ k.l[3];
    return h ^ (h >> 31);
}
}  // namespace

extern "C" int kh_lookup_sorted(const uint64_t* table, size_t lookup_rows, const uint64_t* values, size_t value_stride, size_t max_per_row, uint64_t* out,
                                size_t* bad_row) {
    if (!table || !values || !out || lookup_rows == 0 || max_per_row == 0 || value_stride < lookup_rows) { kh::set_error("kh_lookup_sorted: bad argument"); return KH_E_INVALID; }
    if (bad_row) *bad_row = (size_t)-1;
    const Key* T = (const Key*)table;
    size_t cap = 16; while (cap < 2 * lookup_rows) cap <<= 1;
    const size_t mask = cap - 1;
    std::vector<uint32_t> slot(cap, 0xffffffffu);      // open addressing: table index of the FIRST occurrence of a value
    std::vector<uint32_t> cnt(lookup_rows, 1);         // counts live on first occurrences; a repeated table entry keeps its 1
    std::vector<uint8_t> first(lookup_rows, 0);
    for (size_t i = 0; i < lookup_rows; i++) {
        size_t h = hash(T[i]) & mask;
        while (slot[h] != 0xffffffffu && !same(T[slot[h]], T[i])) h = (h + 1) & mask;
        if (slot[h] == 0xffffffffu) { slot[h] = (uint32_t)i; first[i] = 1; }
    }
    for (size_t s = 0; s < max_per_row; s++) {
        const Key* V = (const Key*)(values + 4 * s * value_stride);
        for (size_t r = 0; r < lookup_rows; r++) {
            size_t h = hash(V[r]) & mask;
            while (slot[h] != 0xffffffffu && !same(T[slot[h]], V[r])) h = (h + 1) & mask;
            if (slot[h] == 0xffffffffu) {               // a looked-up value that is not in the table (constraints.rs:137-141)
                if (bad_row) *bad_row = r;
                kh::set_error("lookup in row %zu (slot %zu): the value is not in the table", r, s);
                return KH_E_INVALID;
            }
            cnt[slot[h]]++;
        }
    }
    // the sorted multiset, table order, cut into max_per_row + 1 columns of lookup_rows values + the snake's shared element
    const size_t L = lookup_rows, W = L + 1;
    Key* O = (Key*)out;
    size_t pos = 0;
    for (size_t i = 0; i < L; i++) {
        const size_t c = first[i] ? cnt[i] : 1;
        for (size_t j = 0; j < c; j++, pos++) O[(pos / L) * W + pos % L] = T[i];
    }
    if (pos != (max_per_row + 1) * L) { kh::set_error("kh_lookup_sorted: %zu values for %zu places", pos, (max_per_row + 1) * L); return KH_E_INVALID; }
    for (size_t k = 0; k < max_per_row; k++) O[k * W + L] = O[(k + 1) * W];
    O[max_per_row * W + L] = O[max_per_row * W + L - 1];
    for (size_t k = 1; k <= max_per_row; k += 2)
        for (size_t a = 0, b = L; a < b; a++, b--) { Key t = O[k * W + a]; O[k * W + a] = O[k * W + b]; O[k * W + b] = t; }
    return KH_OK;
}
