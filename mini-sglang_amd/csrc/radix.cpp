// Native radix prefix tree (host code): the structure walk of the reference's RadixPrefixCache
// (P/kvcache/radix_cache.py:17-236) -- child lookup by first page, first-difference compare, page-aligned split,
// reference counts, LRU eviction over unreferenced leaves -- behind the C ABI (include/msgl_hip.h, msgl_radix_*).
// the reference walks the tree in Python, one dict lookup, one tensor slice and one
// tvm-ffi call (fast_compare_key) per node.
//
// What stays in Python (mini-sglang_amd/radix.py): the per-node VALUE tensors (KV pool slots, device memory) and
// the clock.  Every call that changes the tree reports the node ids involved so that the wrapper can mirror it on the
// values (a split cuts the value tensor at the same position).
//
// Behaviour pinned by the reference, kept here on purpose:
//   * children are an insertion-ordered map keyed by the first page of the child's key (python dict: a replaced
//     key keeps its position, a deleted one loses it); the eviction scan visits them in that order;
//   * eviction = heapq over (timestamp) of unreferenced leaves: heapify / heappop / heappush are CPython's sift
//     procedures verbatim in structure, so that equal timestamps leave in the same order as in the reference;
//   * a split node inherits the timestamp, the walk stamps only fully matched nodes, and returns right after a split.
#include <stdint.h>
#include <string.h>

#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace msgl {
namespace {

struct RadixNode {
  int64_t id = 0;
  RadixNode* parent = nullptr;
  std::vector<int32_t> key;
  int64_t ref = 0;
  int64_t ts = 0;
  // insertion-ordered children: `order` holds the live children with tombstones (nullptr) where one was deleted
  std::vector<RadixNode*> order;
  std::unordered_map<std::string, size_t> index;  // first page (bytes) -> position in `order`
  size_t live = 0;
  bool is_leaf() const { return live == 0; }
};

struct RadixTree {
  int page = 1;
  int64_t next_id = 0;
  int64_t evictable = 0, protected_ = 0;
  RadixNode* root = nullptr;
  std::unordered_map<int64_t, std::unique_ptr<RadixNode>> nodes;

  RadixNode* make(int64_t ts) {
    auto n = std::make_unique<RadixNode>();
    n->id = next_id++;
    n->ts = ts;
    RadixNode* raw = n.get();
    nodes.emplace(raw->id, std::move(n));
    return raw;
  }
  RadixNode* find(int64_t id) const {
    auto it = nodes.find(id);
    return it == nodes.end() ? nullptr : it->second.get();
  }
  std::string page_key(const int32_t* ids) const {
    return std::string(reinterpret_cast<const char*>(ids), sizeof(int32_t) * (size_t)page);
  }
  // parent.children[key_fn(child.key)] = child   (radix_cache.py:39-41)
  void set_parent(RadixNode* child, RadixNode* parent) {
    child->parent = parent;
    const std::string k = page_key(child->key.data());
    auto it = parent->index.find(k);
    if (it != parent->index.end()) {
      parent->order[it->second] = child;  // replaced in place: position kept
    } else {
      parent->index.emplace(k, parent->order.size());
      parent->order.push_back(child);
      ++parent->live;
    }
  }
  void remove_child(RadixNode* parent, RadixNode* child) {
    const std::string k = page_key(child->key.data());
    auto it = parent->index.find(k);
    if (it == parent->index.end()) return;
    parent->order[it->second] = nullptr;
    parent->index.erase(it);
    --parent->live;
    if (parent->order.size() > 32 && parent->live * 2 < parent->order.size()) {  // compact the tombstones
      std::vector<RadixNode*> kept;
      kept.reserve(parent->live);
      for (RadixNode* c : parent->order)
        if (c) kept.push_back(c);
      parent->order.swap(kept);
      for (size_t i = 0; i < parent->order.size(); ++i) parent->index[page_key(parent->order[i]->key.data())] = i;
    }
  }
  // radix_cache.py:65-77
  RadixNode* split_at(RadixNode* node, int64_t pos) {
    RadixNode* parent = node->parent;
    RadixNode* head = make(node->ts);
    head->key.assign(node->key.begin(), node->key.begin() + pos);
    set_parent(head, parent);
    head->ref = node->ref;
    node->key.erase(node->key.begin(), node->key.begin() + pos);
    set_parent(node, head);
    return head;
  }
};

inline int64_t first_diff(const int32_t* a, int64_t na, const int32_t* b, int64_t nb) {
  const int64_t n = na < nb ? na : nb;
  int64_t i = 0;
  for (; i < n; ++i)
    if (a[i] != b[i]) break;
  return i;
}

// CPython heapq on node pointers ordered by timestamp (Lib/heapq.py: _siftdown, _siftup, heapify, heappop, heappush)
inline bool node_lt(const RadixNode* a, const RadixNode* b) { return a->ts < b->ts; }
void sift_down(std::vector<RadixNode*>& h, size_t start, size_t pos) {
  RadixNode* item = h[pos];
  while (pos > start) {
    const size_t pp = (pos - 1) >> 1;
    if (node_lt(item, h[pp])) {
      h[pos] = h[pp];
      pos = pp;
      continue;
    }
    break;
  }
  h[pos] = item;
}
void sift_up(std::vector<RadixNode*>& h, size_t pos) {
  const size_t end = h.size(), start = pos;
  RadixNode* item = h[pos];
  size_t child = 2 * pos + 1;
  while (child < end) {
    const size_t right = child + 1;
    if (right < end && !node_lt(h[child], h[right])) child = right;
    h[pos] = h[child];
    pos = child;
    child = 2 * pos + 1;
  }
  h[pos] = item;
  sift_down(h, start, pos);
}
void heapify(std::vector<RadixNode*>& h) {
  for (size_t i = h.size() / 2; i-- > 0;) sift_up(h, i);
}
RadixNode* heappop(std::vector<RadixNode*>& h) {
  RadixNode* last = h.back();
  h.pop_back();
  if (h.empty()) return last;
  RadixNode* top = h[0];
  h[0] = last;
  sift_up(h, 0);
  return top;
}
void heappush(std::vector<RadixNode*>& h, RadixNode* n) {
  h.push_back(n);
  sift_down(h, 0, h.size() - 1);
}

inline RadixTree* tree_of(void* t) { return static_cast<RadixTree*>(t); }

}  // namespace
}  // namespace msgl

using namespace msgl;

extern "C" int msgl_radix_create(void** tree, int page_size, int64_t now_ns) {
  MSGL_REQUIRE(tree, "radix_create: null pointer");
  MSGL_REQUIRE(page_size >= 1 && page_size <= 4096, "radix_create: page_size %d", page_size);
  auto* t = new RadixTree();
  t->page = page_size;
  t->root = t->make(now_ns);
  t->root->ref = 1;  // the root is always protected (radix_cache.py:108-109)
  *tree = t;
  return MSGL_OK;
}

extern "C" int msgl_radix_destroy(void* tree) {
  delete tree_of(tree);
  return MSGL_OK;
}

// _tree_walk (radix_cache.py:205-230).  out = {node id, matched length, split: head id, split: tail id, split: position}
// (the three split fields are -1 when no node was split).
extern "C" int msgl_radix_walk(void* tree, const int32_t* ids, int64_t n, int64_t now_ns, int64_t* out) {
  MSGL_REQUIRE(tree && out && (ids || n == 0) && n >= 0, "radix_walk: bad arguments");
  RadixTree* t = tree_of(tree);
  RadixNode* node = t->root;
  int64_t prefix = 0;
  out[2] = out[3] = out[4] = -1;
  while (prefix < n) {
    if (n - prefix < t->page) break;  // a shorter remainder has no first page to look up
    auto it = node->index.find(t->page_key(ids + prefix));
    if (it == node->index.end()) break;
    node = node->order[it->second];
    int64_t m = first_diff(node->key.data(), (int64_t)node->key.size(), ids + prefix, n - prefix);
    m -= m % t->page;
    prefix += m;
    if (m != (int64_t)node->key.size()) {
      MSGL_REQUIRE(m > 0, "radix_walk: matched child shares no whole page (corrupt tree)");
      RadixNode* tail = node;
      node = t->split_at(node, m);
      out[2] = node->id;
      out[3] = tail->id;
      out[4] = m;
      break;
    }
    node->ts = now_ns;
  }
  out[0] = node->id;
  out[1] = prefix;
  return MSGL_OK;
}

// the new leaf of insert_prefix (radix_cache.py:140-145): key = the ids beyond the matched prefix
extern "C" int64_t msgl_radix_add_child(void* tree, int64_t parent_id, const int32_t* key, int64_t n, int64_t now_ns) {
  MSGL_REQUIRE(tree && key, "radix_add_child: null pointer");
  RadixTree* t = tree_of(tree);
  RadixNode* parent = t->find(parent_id);
  MSGL_REQUIRE(parent, "radix_add_child: unknown node %lld", (long long)parent_id);
  MSGL_REQUIRE(n >= t->page && n % t->page == 0, "radix_add_child: key of %lld ids is not a whole number of pages",
               (long long)n);
  RadixNode* node = t->make(now_ns);
  node->key.assign(key, key + n);
  t->set_parent(node, parent);
  t->evictable += n;
  return node->id;
}

// lock_handle (radix_cache.py:111-130)
extern "C" int msgl_radix_lock(void* tree, int64_t node_id, int unlock) {
  MSGL_REQUIRE(tree, "radix_lock: null pointer");
  RadixTree* t = tree_of(tree);
  RadixNode* node = t->find(node_id);
  MSGL_REQUIRE(node, "radix_lock: unknown node %lld (evicted?)", (long long)node_id);
  for (; node->parent; node = node->parent) {
    const int64_t len = (int64_t)node->key.size();
    if (unlock) {
      MSGL_REQUIRE(node->ref > 0, "radix_lock: unlock of a node that is not locked");
      if (--node->ref == 0) {
        t->evictable += len;
        t->protected_ -= len;
      }
    } else {
      if (node->ref == 0) {
        t->evictable -= len;
        t->protected_ += len;
      }
      ++node->ref;
    }
  }
  return MSGL_OK;
}

// evict (radix_cache.py:147-175): ids of the evicted nodes in eviction order; returns their count (or < 0)
extern "C" int64_t msgl_radix_evict(void* tree, int64_t size, int64_t* out_ids, int64_t capacity) {
  MSGL_REQUIRE(tree && size >= 0, "radix_evict: bad arguments");
  RadixTree* t = tree_of(tree);
  if (size == 0) return 0;
  MSGL_REQUIRE(size <= t->evictable, "Cannot evict %lld, only %lld is evictable", (long long)size,
               (long long)t->evictable);
  // _collect_leave_nodes_for_evict (radix_cache.py:189-203): stack, children pushed in dict order
  std::vector<RadixNode*> stack{t->root}, heap;
  while (!stack.empty()) {
    RadixNode* n = stack.back();
    stack.pop_back();
    if (n->is_leaf()) {
      if (n->ref == 0) heap.push_back(n);
    } else {
      for (RadixNode* c : n->order)
        if (c) stack.push_back(c);
    }
  }
  heapify(heap);
  int64_t evicted = 0, count = 0;
  while (evicted < size) {
    MSGL_REQUIRE(!heap.empty(), "Cannot evict enough cache, need %lld, only %lld evicted", (long long)size,
                 (long long)evicted);
    RadixNode* n = heappop(heap);
    MSGL_REQUIRE(n->ref == 0 && n->is_leaf() && n->parent, "radix_evict: heap holds a node that must stay");
    MSGL_REQUIRE(count < capacity, "radix_evict: more than %lld nodes to report", (long long)capacity);
    const int64_t len = (int64_t)n->key.size();
    evicted += len;
    out_ids[count++] = n->id;
    t->evictable -= len;
    RadixNode* parent = n->parent;
    t->remove_child(parent, n);
    t->nodes.erase(n->id);
    if (parent->is_leaf() && parent->ref == 0) heappush(heap, parent);
  }
  return count;
}

// node ids from the first node below the root down to `node_id` (RadixCacheHandle.get_matched_indices concatenates
// their values in this order, radix_cache.py:87-94); returns the count
extern "C" int64_t msgl_radix_path(void* tree, int64_t node_id, int64_t* out_ids, int64_t capacity) {
  MSGL_REQUIRE(tree, "radix_path: null pointer");
  RadixTree* t = tree_of(tree);
  RadixNode* node = t->find(node_id);
  MSGL_REQUIRE(node, "radix_path: unknown node %lld (evicted?)", (long long)node_id);
  int64_t depth = 0;
  for (RadixNode* n = node; n->parent; n = n->parent) ++depth;
  MSGL_REQUIRE(depth <= capacity, "radix_path: path of %lld nodes exceeds the buffer", (long long)depth);
  int64_t i = depth;
  for (RadixNode* n = node; n->parent; n = n->parent) out_ids[--i] = n->id;
  return depth;
}

// out = {evictable size, protected size, live nodes (incl. root), key length of `node_id` (or -1)}
extern "C" int msgl_radix_info(void* tree, int64_t node_id, int64_t* out) {
  MSGL_REQUIRE(tree && out, "radix_info: null pointer");
  RadixTree* t = tree_of(tree);
  out[0] = t->evictable;
  out[1] = t->protected_;
  out[2] = (int64_t)t->nodes.size();
  RadixNode* n = node_id >= 0 ? t->find(node_id) : nullptr;
  out[3] = n ? (int64_t)n->key.size() : -1;
  return MSGL_OK;
}

// check_integrity support: recompute the two sizes from the tree and compare every parent / index link
extern "C" int msgl_radix_check(void* tree) {
  MSGL_REQUIRE(tree, "radix_check: null pointer");
  RadixTree* t = tree_of(tree);
  int64_t ev = 0, pr = 0;
  size_t seen = 0;
  std::vector<RadixNode*> stack{t->root};
  while (!stack.empty()) {
    RadixNode* n = stack.back();
    stack.pop_back();
    ++seen;
    if (n->parent) {
      MSGL_REQUIRE(!n->key.empty() && n->key.size() % (size_t)t->page == 0, "radix_check: node %lld has a ragged key",
                   (long long)n->id);
      (n->ref == 0 ? ev : pr) += (int64_t)n->key.size();
      MSGL_REQUIRE(n->ref <= n->parent->ref || !n->parent->parent, "radix_check: node %lld is referenced more than its parent",
                   (long long)n->id);
    }
    size_t live = 0;
    for (size_t i = 0; i < n->order.size(); ++i) {
      RadixNode* c = n->order[i];
      if (!c) continue;
      ++live;
      auto it = n->index.find(t->page_key(c->key.data()));
      MSGL_REQUIRE(c->parent == n && it != n->index.end() && it->second == i, "radix_check: broken child link under %lld",
                   (long long)n->id);
      stack.push_back(c);
    }
    MSGL_REQUIRE(live == n->live && live == n->index.size(), "radix_check: child count of %lld", (long long)n->id);
  }
  MSGL_REQUIRE(seen == t->nodes.size(), "radix_check: %zu reachable nodes of %zu", seen, t->nodes.size());
  MSGL_REQUIRE(ev == t->evictable && pr == t->protected_, "radix_check: sizes (%lld, %lld) but the tree holds (%lld, %lld)",
               (long long)t->evictable, (long long)t->protected_, (long long)ev, (long long)pr);
  return MSGL_OK;
}
