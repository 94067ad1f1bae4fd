// TEST INFRASTRUCTURE -- never part of libmm2amd.so.
//
// Stand-ins for the launchers of the DP kernels whose bodies are gfx950 inline assembly (ksw_gapfill.hip, ksw_stream.hip, ksw_splice.hip) in the
// host build of the product under the wave emulator (libmm2amd_emu.so): the jobs of a launch are computed by the oracle's plain-C restatement,
// results and CIGARs are laid down exactly as the kernels lay them down.  The seeding / sorting / chaining kernels (seed_chain.hip), the index
// build (index_build.hip) and the lane-exact DP kernel (ksw_extd2.hip) run as their own source under the emulator.
#include <hip/hip_runtime.h>
#include <vector>
#include "../../../minimap2_amd/csrc/ksw_dev.hpp"
#include "../../../oracle/oracle.h"

namespace mm2amd {

static void run_jobs(const KswLaunch &L)
{
	std::vector<uint8_t> q, t, jbuf;
	std::vector<uint32_t> cg;
	const KswScoring &sc = L.sc;
	for (int32_t k = 0; k < L.n_jobs; ++k) {
		const KswJob &j = L.jobs[k];
		KswRes &r = L.res[k];
		q.resize(j.qlen > 0 ? j.qlen : 0), t.resize(j.tlen > 0 ? j.tlen : 0);
		for (int i = 0; i < j.qlen; ++i) q[i] = L.qpool[(j.flag & KSWJ_Q_REVERSED) ? j.q_off - i : j.q_off + i];
		for (int i = 0; i < j.tlen; ++i) {
			const uint64_t pos = (j.flag & KSWJ_T_REVERSED) ? j.t_off - i : j.t_off + i;
			t[i] = (j.flag & KSWJ_T_PACKED) ? (uint8_t)(L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : L.tpool[pos];
		}
		ora_ez_t ez;
		cg.resize((size_t)j.qlen + j.tlen + 8);
		if (L.splice) ora_ksw_exts2(j.qlen, q.data(), j.tlen, t.data(), sc.m, sc.mat, sc.q, sc.e, sc.q2, sc.noncan, j.zdrop, j.end_bonus, sc.junc_bonus, sc.junc_pen, j.flag & 0x1fff, nullptr, &ez, cg.data(), (int)cg.size());
		else ora_ksw_extd2(j.qlen, q.data(), j.tlen, t.data(), sc.m, sc.mat, sc.q, sc.e, sc.q2, sc.e2, j.w, j.zdrop, j.end_bonus, j.flag & 0x1fff, &ez, cg.data(), (int)cg.size());
		r.max = ez.max, r.zdropped = ez.zdropped, r.max_q = ez.max_q, r.max_t = ez.max_t, r.mqe = ez.mqe, r.mqe_t = ez.mqe_t;
		r.mte = ez.mte, r.mte_q = ez.mte_q, r.score = ez.score, r.n_cigar = ez.n_cigar, r.reach_end = ez.reach_end;
		r.zd_max = KSW_ZD_NONE, r.zd_t0 = r.zd_t1 = r.zd_q0 = r.zd_q1 = -1; // the host scans the CIGAR itself
		const uint32_t off = __atomic_fetch_add(&L.cigar_cursor[0], (uint32_t)ez.n_cigar, __ATOMIC_RELAXED);
		r.cigar_off = off;
		if ((unsigned long long)off + (unsigned)ez.n_cigar > L.cigar_pool_cap) L.cigar_cursor[1] = 1;
		else for (int i = 0; i < ez.n_cigar; ++i) L.cigar_pool[off + i] = cg[i];
	}
}

void ksw_gapfill_launch(const KswLaunch &L, int, int, void *) { run_jobs(L); }
void ksw_stream_launch(const KswLaunch &L, int, int, void *) { run_jobs(L); }
size_t ksw_stream_slot_bytes(int n_sets) { return (size_t)(n_sets <= 4 ? 512 : 1024) * (size_t)(n_sets * 64) * 4 / 2; }
int ksw_stream_waves(int) { return 4; }
void ksw_splice_launch(const KswLaunch &L, int, int, bool, void *) { run_jobs(L); }

} // namespace mm2amd
