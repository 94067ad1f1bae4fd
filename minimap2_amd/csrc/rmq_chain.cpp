// RMQ-based chaining for the long-join re-chain branch (mg_lchain_rmq, lchain.c:250-368).
//
// The reference keeps the active anchors in an AVL tree ordered by (query position, index) whose nodes carry the
// minimum-priority node of their subtree (krmq.h); a range-minimum query returns one node, and when several nodes
// share the minimum priority the one returned depends on the tree's shape and on the direction in which subtree
// minima are combined.  To give the same chains we maintain the same balanced tree: same insertion/erasure
// rebalancing cases, same "left, then right" combination rule for subtree minima, same query walk.
#include <cassert>
#include <cstring>
#include <vector>
#include "chain_host.hpp"

namespace mm2amd {

namespace {

struct Node {
	int32_t y;            // query position of the anchor
	int64_t i;            // anchor index
	double pri;           // -(f + 0.5 * pen_gap * (x + y)): smaller is better
	Node *c[2];           // children
	Node *best;           // node with the smallest pri in this subtree
	signed char bal;      // height(right) - height(left)
	unsigned size;
};

inline int cmp_node(const Node *a, const Node *b) // lc_elem_cmp, lchain.c:226
{
	return a->y < b->y ? -1 : a->y > b->y ? 1 : (a->i > b->i) - (a->i < b->i);
}
inline bool better(const Node *a, const Node *b) { return a->pri < b->pri; } // lc_elem_lt2
inline unsigned size_of(const Node *p) { return p ? p->size : 0; }

constexpr int MAXD = 64;

struct Tree {
	Node *root = nullptr;

	static void pull_best(Node *p, const Node *q, const Node *r) // krmq_update_min: left subtree first, then right; ties keep the later operand
	{
		p->best = !q || better(p, q->best) ? p : q->best;
		p->best = !r || better(p->best, r->best) ? p->best : r->best;
	}
	// single rotation: (a,(b,c)q)p => ((a,b)p,c)q  ; dir=0 rotates to the left
	static Node *rot1(Node *p, int dir)
	{
		const int opp = 1 - dir;
		Node *q = p->c[opp], *s = p->best;
		const unsigned size_p = p->size;
		p->size -= q->size - size_of(q->c[dir]);
		q->size = size_p;
		pull_best(p, p->c[dir], q->c[dir]);
		q->best = s;
		p->c[opp] = q->c[dir];
		q->c[dir] = p;
		return q;
	}
	// double rotation: (a,((b,c)r,d)q)p => ((a,b)p,(c,d)q)r
	static Node *rot2(Node *p, int dir)
	{
		const int opp = 1 - dir;
		Node *q = p->c[opp], *r = q->c[dir], *s = p->best;
		const unsigned size_x_dir = size_of(r->c[dir]);
		r->size = p->size;
		p->size -= q->size - size_x_dir;
		q->size -= size_x_dir + 1;
		pull_best(p, p->c[dir], r->c[dir]);
		pull_best(q, q->c[opp], r->c[opp]);
		r->best = s;
		p->c[opp] = r->c[dir];
		r->c[dir] = p;
		q->c[dir] = r->c[opp];
		r->c[opp] = q;
		const int b1 = dir == 0 ? +1 : -1;
		if (r->bal == b1) q->bal = 0, p->bal = (signed char)-b1;
		else if (r->bal == 0) q->bal = p->bal = 0;
		else q->bal = (signed char)b1, p->bal = 0;
		r->bal = 0;
		return r;
	}
	void insert(Node *x)
	{
		unsigned char stack[MAXD];
		Node *path[MAXD];
		Node *bp = root, *bq = nullptr, *p, *q, *r = nullptr;
		int which = 0, top = 0, path_len = 0;
		for (p = bp, q = bq; p; q = p, p = p->c[which]) {
			const int cmp = cmp_node(x, p);
			if (cmp == 0) return; // keys are unique here
			if (p->bal != 0) bq = q, bp = p, top = 0;
			stack[top++] = which = (cmp > 0);
			path[path_len++] = p;
		}
		x->bal = 0, x->size = 1, x->c[0] = x->c[1] = nullptr, x->best = x;
		if (q == nullptr) root = x;
		else q->c[which] = x;
		if (bp == nullptr) return;
		for (int k = 0; k < path_len; ++k) ++path[k]->size;
		for (int k = path_len - 1; k >= 0; --k) {
			pull_best(path[k], path[k]->c[0], path[k]->c[1]);
			if (path[k]->best != x) break;
		}
		for (p = bp, top = 0; p != x; p = p->c[stack[top]], ++top)
			if (stack[top] == 0) --p->bal; else ++p->bal;
		if (bp->bal > -2 && bp->bal < 2) return;
		which = (bp->bal < 0);
		const int b1 = which == 0 ? +1 : -1;
		q = bp->c[1 - which];
		if (q->bal == b1) {
			r = rot1(bp, which);
			q->bal = bp->bal = 0;
		} else r = rot2(bp, which);
		if (bq == nullptr) root = r;
		else bq->c[bp != bq->c[0]] = r;
	}
	Node *find(const Node *x) const
	{
		Node *p = root;
		while (p) {
			const int cmp = cmp_node(x, p);
			if (cmp < 0) p = p->c[0];
			else if (cmp > 0) p = p->c[1];
			else break;
		}
		return p;
	}
	Node *erase(const Node *x)
	{
		Node *p, *path[MAXD], fake;
		unsigned char dir[MAXD];
		int d = 0, cmp;
		fake = *root, fake.c[0] = root, fake.c[1] = nullptr;
		for (cmp = -1, p = &fake; cmp; cmp = cmp_node(x, p)) {
			const int which = (cmp > 0);
			dir[d] = (unsigned char)which;
			path[d++] = p;
			p = p->c[which];
			if (p == nullptr) return nullptr;
		}
		for (int k = 1; k < d; ++k) --path[k]->size;
		if (p->c[1] == nullptr) {
			path[d - 1]->c[dir[d - 1]] = p->c[0];
		} else {
			Node *q = p->c[1];
			if (q->c[0] == nullptr) {
				q->c[0] = p->c[0];
				q->bal = p->bal;
				path[d - 1]->c[dir[d - 1]] = q;
				path[d] = q, dir[d++] = 1;
				q->size = p->size - 1;
			} else {
				Node *r;
				const int e = d++;
				for (;;) {
					dir[d] = 0;
					path[d++] = q;
					r = q->c[0];
					if (r->c[0] == nullptr) break;
					q = r;
				}
				r->c[0] = p->c[0];
				q->c[0] = r->c[1];
				r->c[1] = p->c[1];
				r->bal = p->bal;
				path[e - 1]->c[dir[e - 1]] = r;
				path[e] = r, dir[e] = 1;
				for (int k = e + 1; k < d; ++k) --path[k]->size;
				r->size = p->size - 1;
			}
		}
		for (int k = d - 1; k >= 0; --k) pull_best(path[k], path[k]->c[0], path[k]->c[1]);
		while (--d > 0) {
			Node *q = path[d];
			int b1 = 1, b2 = 2;
			const int which = dir[d], other = 1 - which;
			if (which) b1 = -b1, b2 = -b2;
			q->bal = (signed char)(q->bal + b1);
			if (q->bal == b1) break;
			else if (q->bal == b2) {
				Node *r = q->c[other];
				if (r->bal == -b1) {
					path[d - 1]->c[dir[d - 1]] = rot2(q, which);
				} else {
					path[d - 1]->c[dir[d - 1]] = rot1(q, which);
					if (r->bal == 0) {
						r->bal = (signed char)-b1;
						q->bal = (signed char)b1;
						break;
					} else r->bal = q->bal = 0;
				}
			}
		}
		root = fake.c[0];
		return p;
	}
	// minimum-priority node with lo <= key <= up (closed interval); krmq_rmq
	const Node *range_min(const Node *lo, const Node *up) const
	{
		const Node *p = root, *path[2][MAXD], *min;
		int plen[2] = {0, 0}, pcmp[2][MAXD], i, cmp, lca;
		if (root == nullptr) return nullptr;
		while (p) {
			cmp = cmp_node(lo, p);
			path[0][plen[0]] = p, pcmp[0][plen[0]++] = cmp;
			if (cmp < 0) p = p->c[0]; else if (cmp > 0) p = p->c[1]; else break;
		}
		p = root;
		while (p) {
			cmp = cmp_node(up, p);
			path[1][plen[1]] = p, pcmp[1][plen[1]++] = cmp;
			if (cmp < 0) p = p->c[0]; else if (cmp > 0) p = p->c[1]; else break;
		}
		for (i = 0; i < plen[0] && i < plen[1]; ++i)
			if (path[0][i] == path[1][i] && pcmp[0][i] <= 0 && pcmp[1][i] >= 0) break;
		if (i == plen[0] || i == plen[1]) return nullptr;
		lca = i, min = path[0][lca];
		for (i = lca + 1; i < plen[0]; ++i) {
			if (pcmp[0][i] <= 0) {
				if (better(path[0][i], min)) min = path[0][i];
				if (path[0][i]->c[1] && better(path[0][i]->c[1]->best, min)) min = path[0][i]->c[1]->best;
			}
		}
		for (i = lca + 1; i < plen[1]; ++i) {
			if (pcmp[1][i] >= 0) {
				if (better(path[1][i], min)) min = path[1][i];
				if (path[1][i]->c[0] && better(path[1][i]->c[0]->best, min)) min = path[1][i]->c[0]->best;
			}
		}
		return min;
	}
	// largest node <= x (krmq_interval's lower bound)
	const Node *floor(const Node *x) const
	{
		const Node *p = root, *l = nullptr;
		while (p) {
			const int cmp = cmp_node(x, p);
			if (cmp < 0) p = p->c[0];
			else if (cmp > 0) l = p, p = p->c[1];
			else { l = p; break; }
		}
		return l;
	}
};

// in-order iterator that can step backwards (krmq_itr_find + krmq_itr_prev)
struct Iter {
	const Node *stack[MAXD];
	int top = -1;
	bool seek(const Node *root, const Node *x)
	{
		const Node *p = root;
		top = -1;
		while (p) {
			stack[++top] = p;
			const int cmp = cmp_node(x, p);
			if (cmp < 0) p = p->c[0]; else if (cmp > 0) p = p->c[1]; else break;
		}
		return p != nullptr;
	}
	const Node *at() const { return top < 0 ? nullptr : stack[top]; }
	bool prev()
	{
		if (top < 0) return false;
		const Node *p = stack[top]->c[0];
		if (p) {
			for (; p; p = p->c[1]) stack[++top] = p;
			return true;
		}
		do { p = stack[top--]; } while (top >= 0 && p == stack[top]->c[0]);
		return top >= 0;
	}
};

struct Pool { // fixed-capacity node pool with a free list (kmp_*_rmq)
	std::vector<Node> mem;
	std::vector<Node *> free_list;
	size_t used = 0;
	explicit Pool(size_t cap) : mem(cap) {}
	Node *get() { if (!free_list.empty()) { Node *p = free_list.back(); free_list.pop_back(); return p; } return &mem[used++]; }
	void put(Node *p) { free_list.push_back(p); }
};

// comput_sc_simple, lchain.c:229-248
inline int32_t simple_score(const Anchor &ai, const Anchor &aj, float pen_gap, float pen_skip, int32_t *exact, int32_t *width)
{
	const int32_t dq = (int32_t)ai.y - (int32_t)aj.y, dr = (int32_t)(ai.x - aj.x);
	const int32_t dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq, span = (int32_t)(aj.y >> 32 & 0xff);
	int32_t sc = span < dg ? span : dg;
	*width = dd;
	if (exact) *exact = (dd == 0 && dg <= span);
	if (dd || dq > span) {
		const float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		const float lg = dd >= 1 ? fast_log2((float)(dd + 1)) : 0.0f;
		sc -= (int)(lin + .5f * lg);
	}
	return sc;
}

} // namespace

void chain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
               float pen_gap, float pen_skip, int64_t n, const Anchor *a, std::vector<uint64_t> &u, std::vector<Anchor> &out,
               ChainScratch &sc)
{
	u.clear(); out.clear();
	if (n == 0) return;
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner < 0) max_dist_inner = 0;
	if (max_dist_inner > max_dist) max_dist_inner = max_dist;
	std::vector<int32_t> f(n), p(n), t(n, 0);
	Tree outer, inner;
	Pool pool((size_t)n * 2 + 4);
	int64_t i0 = 0, st = 0, st_inner = 0;
	for (int64_t i = 0; i < n; ++i) {
		int64_t max_j = -1;
		const int32_t q_span = (int32_t)(a[i].y >> 32 & 0xff);
		int32_t max_f = q_span;
		Node key, lo, hi;
		if (i0 < i && a[i0].x != a[i].x) { // anchors with a smaller target coordinate become candidates
			for (int64_t j = i0; j < i; ++j) {
				Node *q = pool.get();
				q->y = (int32_t)a[j].y, q->i = j, q->pri = -(f[j] + 0.5 * pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y));
				outer.insert(q);
				if (max_dist_inner > 0) {
					Node *r = pool.get();
					r->y = q->y, r->i = q->i, r->pri = q->pri;
					inner.insert(r);
				}
			}
			i0 = i;
		}
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + max_dist || (int)size_of(outer.root) > cap_rmq_size)) {
			key.y = (int32_t)a[st].y, key.i = st;
			if (outer.root && outer.find(&key)) pool.put(outer.erase(&key));
			++st;
		}
		if (max_dist_inner > 0) {
			while (st_inner < i && (a[i].x >> 32 != a[st_inner].x >> 32 || a[i].x > a[st_inner].x + max_dist_inner || (int)size_of(inner.root) > cap_rmq_size)) {
				key.y = (int32_t)a[st_inner].y, key.i = st_inner;
				if (inner.root && inner.find(&key)) pool.put(inner.erase(&key));
				++st_inner;
			}
		}
		lo.i = INT32_MAX, lo.y = (int32_t)a[i].y - max_dist;
		hi.i = 0, hi.y = (int32_t)a[i].y;
		if (const Node *q = outer.range_min(&lo, &hi)) {
			int32_t s, exact, width, n_skip = 0;
			int64_t j = q->i;
			s = f[j] + simple_score(a[i], a[j], pen_gap, pen_skip, &exact, &width);
			if (width <= bw && s > max_f) max_f = s, max_j = j;
			if (!exact && inner.root && (int32_t)a[i].y > 0) { // exhaustive look at the close neighbourhood
				key.y = (int32_t)a[i].y - 1, key.i = n;
				if (const Node *lower = inner.floor(&key)) {
					Iter it;
					it.seek(inner.root, lower);
					const Node *c;
					while ((c = it.at()) != nullptr) {
						if (c->y < (int32_t)a[i].y - max_dist_inner) break;
						j = c->i;
						s = f[j] + simple_score(a[i], a[j], pen_gap, pen_skip, nullptr, &width);
						if (width <= bw) {
							if (s > max_f) {
								max_f = s, max_j = j;
								if (n_skip > 0) --n_skip;
							} else if (t[j] == (int32_t)i) {
								if (++n_skip > max_chn_skip) break;
							}
							if (p[j] >= 0) t[p[j]] = (int32_t)i;
						}
						if (!it.prev()) break;
					}
				}
			}
		}
		f[i] = max_f, p[i] = (int32_t)max_j;
	}
	chain_backtrack_compact(n, a, f.data(), p.data(), min_cnt, min_sc, bw, u, out, sc);
}

} // namespace mm2amd
