// RMQ-based chaining (mg_lchain_rmq, lchain.c:250-368): the primary chainer of the MM_F_RMQ presets (asm5/10/20, lr:hqae) and
// the long-join re-chain of map-ont / map-hifi (map.c:283-292), on the host.
//
// What the reference computes.  Anchors are sorted by target coordinate; while anchor i is scored, the anchors of a sliding index
// window [st, i0) -- everything already scored whose target coordinate is smaller and at most max_dist away -- are candidates.
// Two look-ups per anchor:
//   (1) among the window's anchors with query coordinate in (y_i - max_dist, y_i): the one with the smallest priority
//       -(f + 0.5 * pen_gap * (x + y));
//   (2) when that one is not an exact diagonal extension: the anchors of a second, narrower window (max_dist_inner) in
//       descending (query coordinate, index) order, scored one by one with the skip rule of the chaining DP.
// Look-up (2) depends only on the ORDER of the keys, so any ordered container gives the reference's answer: a sorted vector here.
// Look-up (1) is a range-minimum query, and its answer is unique only while the minimum is: the reference keeps the window in an
// AVL tree whose nodes remember the minimum-priority node of their subtree (krmq.h), and with equal priorities (two anchors
// with the same chain score on the same anti-diagonal x + y: repeats produce them) the node it returns depends on that tree's
// shape and on how every earlier insertion, removal and rotation happened to break the tie.  To return the same anchor we keep
// a tree that evolves through the same states -- see TieExactMinTree below for the rules that matter and where they come from.
#include <algorithm>
#include <cstring>
#include <vector>
#include "chain_host.hpp"

namespace mm2amd {

namespace {

// ---------------------------------------------------------------------------------------------------------
// Height-balanced search tree over (y, idx) with a per-subtree "best" (smallest-priority) node, kept in an index arena.
//
// Behaviour that must agree with krmq.h for the range minimum to break ties the same way:
//   * shape: textbook AVL insertion; removal replaces a node that has a right subtree by its in-order successor
//     (krmq.h:262-284) and rebalances bottom-up (:287-307).  Any implementation of these rules passes through the same shapes.
//   * best-node bookkeeping is NOT a function of the shape.  combine(node, first, second) takes the node itself unless
//     first's best is at least as good, then second's best unless the current pick is strictly better -- the later operand
//     wins ties (krmq_update_min, :154-157).  It is called with (left, right) after an insertion -- upwards from the new node,
//     stopping at the first ancestor whose best is not the new node (:226-229) -- and for every ancestor after a removal (:285-286);
//     a rotation recombines only the node(s) that moved down, with the operands in the order (child that stays, child that is
//     handed over) (:165,179-180), and the new subtree root INHERITS the old root's best without a recombination (:166,181).
//   * the query visits the split node, then the lower boundary path top-down (node, then its right child's best), then the upper
//     boundary path (node, then its left child's best), and replaces its pick only by a strictly better node (:134-149).
// ---------------------------------------------------------------------------------------------------------
class TieExactMinTree {
public:
	explicit TieExactMinTree(size_t capacity) { pool_.reserve(capacity + 1); pool_.emplace_back(); } // slot 0 is "no node"
	uint32_t size() const { return pool_[root_].count; }
	bool empty() const { return root_ == 0; }

	void insert(int32_t y, int64_t idx, double pri)
	{
		const int32_t fresh = alloc(y, idx, pri);
		trail_.clear();
		for (int32_t cur = root_; cur;) {
			const int side = key_less(cur, y, idx) ? 1 : 0; // keys are unique: (y, idx) with idx the anchor's index
			trail_.push_back({ cur, side });
			cur = pool_[cur].kid[side];
		}
		if (trail_.empty()) { root_ = fresh; return; }
		pool_[trail_.back().node].kid[trail_.back().side] = fresh;
		for (const Step &s : trail_) ++pool_[s.node].count;
		for (size_t k = trail_.size(); k-- > 0;) { // the new node announces itself upwards while it is the best of the subtree
			combine(trail_[k].node, pool_[trail_[k].node].kid[0], pool_[trail_[k].node].kid[1]);
			if (pool_[trail_[k].node].best != fresh) break;
		}
		for (size_t k = trail_.size(); k-- > 0;) { // retrace: the subtree below trail_[k] on its `side` grew by one level
			Node &a = pool_[trail_[k].node];
			a.tilt += trail_[k].side ? 1 : -1;
			if (a.tilt == 0) break;
			if (a.tilt == 1 || a.tilt == -1) continue;
			const int heavy = trail_[k].side;
			const int32_t sub = pool_[a.kid[heavy]].tilt == (heavy ? 1 : -1) ? rotate_once(trail_[k].node, 1 - heavy, true) : rotate_twice(trail_[k].node, 1 - heavy);
			relink(k, sub);
			break;
		}
	}

	// removes (y, idx) if present
	bool erase(int32_t y, int64_t idx)
	{
		trail_.clear();
		int32_t cur = root_;
		while (cur && !key_equal(cur, y, idx)) {
			const int side = key_less(cur, y, idx) ? 1 : 0;
			trail_.push_back({ cur, side });
			cur = pool_[cur].kid[side];
		}
		if (!cur) return false;
		for (const Step &s : trail_) --pool_[s.node].count;
		const int32_t gone = cur;
		const size_t at = trail_.size(); // position of the removed node in the trail
		if (pool_[gone].kid[1] == 0) {
			relink(at, pool_[gone].kid[0]);
		} else {
			int32_t succ = pool_[gone].kid[1];
			if (pool_[succ].kid[0] == 0) { // the right child is the successor: it moves up with its own right subtree
				pool_[succ].kid[0] = pool_[gone].kid[0];
				pool_[succ].tilt = pool_[gone].tilt;
				pool_[succ].count = pool_[gone].count - 1;
				relink(at, succ);
				trail_.push_back({ succ, 1 });
			} else { // the successor is the leftmost node of the right subtree: it takes the removed node's place
				trail_.push_back({ 0, 1 }); // placeholder for the successor
				int32_t parent = succ;
				for (;;) {
					trail_.push_back({ parent, 0 });
					succ = pool_[parent].kid[0];
					if (pool_[succ].kid[0] == 0) break;
					parent = succ;
				}
				pool_[parent].kid[0] = pool_[succ].kid[1];
				pool_[succ].kid[0] = pool_[gone].kid[0], pool_[succ].kid[1] = pool_[gone].kid[1];
				pool_[succ].tilt = pool_[gone].tilt;
				for (size_t k = at + 1; k < trail_.size(); ++k) --pool_[trail_[k].node].count;
				pool_[succ].count = pool_[gone].count - 1;
				trail_[at].node = succ;
				relink(at, succ);
			}
		}
		for (size_t k = trail_.size(); k-- > 0;) combine(trail_[k].node, pool_[trail_[k].node].kid[0], pool_[trail_[k].node].kid[1]);
		for (size_t k = trail_.size(); k-- > 0;) { // retrace: the subtree below trail_[k] on its `side` lost one level
			const int32_t a = trail_[k].node;
			const int side = trail_[k].side, away = side ? -1 : 1; // the tilt moves away from the side that shrank
			pool_[a].tilt += away;
			if (pool_[a].tilt == away) break;      // it was balanced: its height is unchanged
			if (pool_[a].tilt == 0) continue;      // it leaned to the shrunk side: one level shorter now
			const int32_t other = pool_[a].kid[1 - side];
			if (pool_[other].tilt == -away) { relink(k, rotate_twice(a, side)); continue; }
			const bool level = pool_[other].tilt == 0;
			relink(k, rotate_once(a, side, !level));
			if (level) { pool_[other].tilt = -away, pool_[a].tilt = away; break; } // the rotated subtree kept its height
		}
		recycle(gone);
		return true;
	}

	// minimum-priority node with lo < key <= hi as the reference delimits it: keys (y_lo, INT32_MAX) .. (y_hi, 0), closed
	int64_t range_min(int32_t y_lo, int32_t y_hi) const
	{
		const int64_t i_lo = INT32_MAX, i_hi = 0;
		int32_t split = root_;
		while (split) { // the first node inside the interval on the way down
			if (order(y_lo, i_lo, split) > 0) split = pool_[split].kid[1];
			else if (order(y_hi, i_hi, split) < 0) split = pool_[split].kid[0];
			else break;
		}
		if (!split) return -1;
		int32_t pick = split;
		auto offer = [&](int32_t c) { if (c && pool_[c].pri < pool_[pick].pri) pick = c; };
		if (order(y_lo, i_lo, split) != 0) // lower boundary: nodes at or above the bound count, together with everything to their right
			for (int32_t cur = pool_[split].kid[0]; cur;) {
				const int c = order(y_lo, i_lo, cur);
				if (c <= 0) { offer(cur); offer(pool_[pool_[cur].kid[1]].best_or_nil(pool_[cur].kid[1])); }
				if (c == 0) break;
				cur = pool_[cur].kid[c < 0 ? 0 : 1];
			}
		if (order(y_hi, i_hi, split) != 0) // upper boundary, mirrored
			for (int32_t cur = pool_[split].kid[1]; cur;) {
				const int c = order(y_hi, i_hi, cur);
				if (c >= 0) { offer(cur); offer(pool_[pool_[cur].kid[0]].best_or_nil(pool_[cur].kid[0])); }
				if (c == 0) break;
				cur = pool_[cur].kid[c < 0 ? 0 : 1];
			}
		return pool_[pick].idx;
	}

private:
	struct Node {
		int32_t y = 0, kid[2] = { 0, 0 }, best = 0;
		int32_t tilt = 0;       // height(right) - height(left)
		uint32_t count = 0;     // nodes in the subtree
		int64_t idx = 0;
		double pri = 0;
		int32_t best_or_nil(int32_t self) const { return self ? best : 0; }
	};
	struct Step { int32_t node; int side; };
	std::vector<Node> pool_;
	std::vector<int32_t> spare_;
	mutable std::vector<Step> trail_;
	int32_t root_ = 0;

	int32_t alloc(int32_t y, int64_t idx, double pri)
	{
		int32_t id;
		if (!spare_.empty()) id = spare_.back(), spare_.pop_back();
		else id = (int32_t)pool_.size(), pool_.emplace_back();
		Node &n = pool_[id];
		n = Node();
		n.y = y, n.idx = idx, n.pri = pri, n.best = id, n.count = 1;
		return id;
	}
	void recycle(int32_t id) { spare_.push_back(id); }
	bool key_less(int32_t n, int32_t y, int64_t idx) const { return pool_[n].y < y || (pool_[n].y == y && pool_[n].idx < idx); } // node key < (y, idx)
	bool key_equal(int32_t n, int32_t y, int64_t idx) const { return pool_[n].y == y && pool_[n].idx == idx; }
	int order(int32_t y, int64_t idx, int32_t n) const { return y < pool_[n].y ? -1 : y > pool_[n].y ? 1 : (idx > pool_[n].idx) - (idx < pool_[n].idx); } // (y, idx) vs node key
	// the node hanging where trail_[k] hangs (k == trail_.size(): below the last step) is replaced by `sub`
	void relink(size_t k, int32_t sub)
	{
		if (k == 0) root_ = sub;
		else pool_[trail_[k - 1].node].kid[trail_[k - 1].side] = sub;
	}
	void combine(int32_t n, int32_t first, int32_t second)
	{
		int32_t pick = n;
		if (first && !(pool_[n].pri < pool_[pool_[first].best].pri)) pick = pool_[first].best;
		if (second && !(pool_[pick].pri < pool_[pool_[second].best].pri)) pick = pool_[second].best;
		pool_[n].best = pick;
	}
	// `top` sinks towards `dir`, its child on the other side comes up.  flatten: both end up balanced (the caller fixes the tilts otherwise)
	int32_t rotate_once(int32_t top, int dir, bool flatten)
	{
		const int32_t up = pool_[top].kid[1 - dir], whole = pool_[top].best;
		const uint32_t all = pool_[top].count;
		pool_[top].count -= pool_[up].count - pool_[pool_[up].kid[dir]].count;
		pool_[up].count = all;
		combine(top, pool_[top].kid[dir], pool_[up].kid[dir]);
		pool_[up].best = whole;
		pool_[top].kid[1 - dir] = pool_[up].kid[dir];
		pool_[up].kid[dir] = top;
		if (flatten) pool_[up].tilt = pool_[top].tilt = 0;
		return up;
	}
	// the grandchild between `top` and its child on the far side comes up; `top` sinks towards `dir`
	int32_t rotate_twice(int32_t top, int dir)
	{
		const int far = 1 - dir;
		const int32_t mid = pool_[top].kid[far], up = pool_[mid].kid[dir], whole = pool_[top].best;
		const uint32_t handed = pool_[pool_[up].kid[dir]].count;
		pool_[up].count = pool_[top].count;
		pool_[top].count -= pool_[mid].count - handed;
		pool_[mid].count -= handed + 1;
		combine(top, pool_[top].kid[dir], pool_[up].kid[dir]);
		combine(mid, pool_[mid].kid[far], pool_[up].kid[far]);
		pool_[up].best = whole;
		pool_[top].kid[far] = pool_[up].kid[dir], pool_[up].kid[dir] = top;
		pool_[mid].kid[dir] = pool_[up].kid[far], pool_[up].kid[far] = mid;
		const int lean = dir == 0 ? 1 : -1; // the tilt `up` had decides which of the two takes the shorter half
		if (pool_[up].tilt == lean) pool_[mid].tilt = 0, pool_[top].tilt = -lean;
		else if (pool_[up].tilt == 0) pool_[mid].tilt = pool_[top].tilt = 0;
		else pool_[mid].tilt = lean, pool_[top].tilt = 0;
		pool_[up].tilt = 0;
		return up;
	}
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

inline uint64_t near_key(int32_t y, int64_t idx) { return (uint64_t)((uint32_t)y ^ 0x80000000u) << 32 | (uint64_t)(uint32_t)idx; } // sorts like (y, idx)
inline int32_t near_y(uint64_t k) { return (int32_t)((uint32_t)(k >> 32) ^ 0x80000000u); }

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
	TieExactMinTree far((size_t)std::min<int64_t>(n, (int64_t)cap_rmq_size + 2));
	std::vector<uint64_t> near; // the narrow window's keys, ascending
	int64_t i0 = 0, st = 0, st_near = 0;
	for (int64_t i = 0; i < n; ++i) {
		int64_t max_j = -1;
		const int32_t y_i = (int32_t)a[i].y;
		int32_t max_f = (int32_t)(a[i].y >> 32 & 0xff);
		if (i0 < i && a[i0].x != a[i].x) { // anchors with a smaller target coordinate become candidates (lchain.c:285-298)
			for (int64_t j = i0; j < i; ++j) {
				far.insert((int32_t)a[j].y, j, -(f[j] + 0.5 * pen_gap * ((int32_t)a[j].x + (int32_t)a[j].y)));
				if (max_dist_inner > 0) { const uint64_t k = near_key((int32_t)a[j].y, j); near.insert(std::lower_bound(near.begin(), near.end(), k), k); }
			}
			i0 = i;
		}
		// candidates out of reach -- another sequence, too far back, or more of them than the cap -- leave in index order (:300-318)
		for (; st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + max_dist || (int64_t)far.size() > cap_rmq_size); ++st)
			far.erase((int32_t)a[st].y, st);
		if (max_dist_inner > 0)
			for (; st_near < i && (a[i].x >> 32 != a[st_near].x >> 32 || a[i].x > a[st_near].x + max_dist_inner || (int64_t)near.size() > cap_rmq_size); ++st_near) {
				const uint64_t k = near_key((int32_t)a[st_near].y, st_near);
				const auto it = std::lower_bound(near.begin(), near.end(), k);
				if (it != near.end() && *it == k) near.erase(it);
			}
		const int64_t best = far.range_min(y_i - max_dist, y_i); // :320-322
		if (best >= 0) {
			int32_t exact, width, n_skip = 0;
			int32_t s = f[best] + simple_score(a[i], a[best], pen_gap, pen_skip, &exact, &width);
			if (width <= bw && s > max_f) max_f = s, max_j = best;
			if (!exact && !near.empty() && y_i > 0) { // the close neighbourhood one by one, nearest query coordinate first (:328-354)
				size_t k = (size_t)(std::upper_bound(near.begin(), near.end(), near_key(y_i - 1, n)) - near.begin());
				while (k-- > 0) {
					if (near_y(near[k]) < y_i - max_dist_inner) break;
					const int64_t j = (int64_t)(uint32_t)near[k];
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
				}
			}
		}
		f[i] = max_f, p[i] = (int32_t)max_j;
	}
	chain_backtrack_compact(n, a, f.data(), p.data(), min_cnt, min_sc, bw, u, out, sc);
}

} // namespace mm2amd
