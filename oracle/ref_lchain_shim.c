/* TEST INFRASTRUCTURE ONLY (oracle/Makefile -> oracle/_ref/libreflchain.so; never linked into the product).
 * The reference's mg_chain_bk_end (lchain.c:9-25) is `static`: this translation unit compiles the reference's lchain.c where it lies (-I$(REF); nothing is
 * copied) and adds one entry point that calls it, so that chain_host.cpp's chain_cut can be pinned to it on its own (tests/cpucheck/region_rules_test.cpp).
 * The external functions lchain.c defines are renamed so that the shim can sit beside the reference library. */
#define mg_chain_backtrack refshim_mg_chain_backtrack
#define mg_lchain_dp refshim_mg_lchain_dp
#define mg_lchain_rmq refshim_mg_lchain_rmq
#include <lchain.c> /* (angle brackets: the reference's file on the -I path, not the restatement of the same name beside this shim) */

/* z_x = the chain end's score, z_y = its anchor; f, p, t as in mg_chain_backtrack; t must come back unchanged */
int64_t refshim_chain_bk_end(int32_t max_drop, int32_t z_x, int64_t z_y, const int32_t *f, const int64_t *p, int32_t *t)
{
	mm128_t z;
	z.x = (uint64_t)(uint32_t)z_x, z.y = (uint64_t)z_y;
	return mg_chain_bk_end(max_drop, &z, f, p, t, 0);
}
