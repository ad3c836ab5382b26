/*
 * ref_rnd.cpp — builds the REFERENCE's own host random-number header as a checker.  TEST INFRASTRUCTURE ONLY.
 *
 * Caitlyn/Rnd.h (the PCG hash behind `randomVector`, Scene.h:1208, and the xorshift `randf`) is plain standard C++ — the one
 * host header of the reference besides stb_image that needs neither glm nor GL — so it compiles here with g++ alone.  This
 * translation unit only includes it FROM WHERE IT LIES (-I/root/reference/Caitlyn; nothing of it is copied into this
 * repository) and exports C entry points around its functions; the result goes to oracle/_ref/librndref.so (git-ignored,
 * build container only).  tests/golden/make_ref_rnd_fixture.py records its outputs; tests/test_host.py holds crt_randf2 /
 * crt_pcg_hash to them.
 */
#include <cstdint>
#include "Rnd.h"

extern "C" {
void ref_rnd_set_state(uint32_t s) { s_RndState = s; }
uint32_t ref_rnd_state() { return s_RndState; }
float ref_randf2() { return randf2(); }
float ref_randf() { return randf(); }
uint32_t ref_pcg_hash(uint32_t x) { return PCG_Hash(x); }
}
