// Host RNG that produces the per-frame `randomVector` (Caitlyn/Rnd.h:7-40, used at
// Caitlyn/Scene.h:1208-1210).  State starts at 1; each draw hashes the state in place.
#pragma once
#include <cstdint>

namespace crt {

inline uint32_t pcg_hash(uint32_t input) {            // Rnd.h:21-26
    uint32_t state = input * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

struct Rnd {
    uint32_t state = 1;                               // Rnd.h:7
    float randf2() {                                  // Rnd.h:36-40
        state = pcg_hash(state);
        // Rnd.h:8: imax = 1.0f / UINT32_MAX (the constant rounds to 2^-32); uint -> float
        // conversion rounds to nearest.
        return (float)state * (1.0f / (float)UINT32_MAX);
    }
};

}  // namespace crt
