/*
    autodiff_impl.h -- definition of Tape<Value> (declared in include/enoki/autodiff.h)

    The tape is a host-side DAG: nodes are numbered in creation order (so ascending index order is a
    topological order), every node stores its incoming edges {source, weight | special} plus the list
    of nodes that consume it.  backward() visits the scheduled nodes in DESCENDING index order and
    pushes gradients along the edges, forward() does the mirror image in ascending order.  Each push
    is one array operation on the wrapped type -- for HIPArray<float> exactly one kernel launch:

        source.grad  = safe_mul(w, target.grad)                     first contribution
        source.grad  = safe_fmadd(w, target.grad, source.grad)      further contributions
        source.grad (+)= hsum(safe_mul(w, target.grad))              scalar source, vector edge

    Behavioural reference: src/autodiff/autodiff.cpp (append* 266-331 and 610-679, refcounts 681-774,
    set_gradient 822-836, backward 838-910, forward 912-988, simplify_graph 990-1074, specials
    354-608).  Include this header in exactly one translation unit per value type and instantiate
    `template struct enoki::Tape<T>;` there.
*/
#pragma once

#include <enoki/autodiff.h>

#include <algorithm>
#include <iomanip>
#include <iostream>
#include <set>
#include <sstream>
#include <unordered_map>

namespace enoki {

/// Upper bound on edges created by eliminating one vertex during simplification (autodiff.cpp:33)
#if !defined(ENOKI_AUTODIFF_MAX_SIMPLIFICATION_COST)
#  define ENOKI_AUTODIFF_MAX_SIMPLIFICATION_COST 10
#endif

template <typename Value> struct Tape<Value>::Special {
    virtual void backward(Detail *, Index /* target */, const Edge &) const {
        throw std::runtime_error("Tape::Special::backward(): not implemented");
    }
    virtual void forward(Detail *, Index /* target */, const Edge &) const {
        throw std::runtime_error("Tape::Special::forward(): not implemented");
    }
    /// The adjoint of a (non-permuting) gather as data -- index array, mask, size of the gathered-from array -- so that
    /// backward() can batch the scatter_adds of gathers that share their index array and fuse pending edge products
    /// into them.  false: not such a gather.
    struct GatherAdjoint {
        const Offset *offset = nullptr;
        const Mask *mask = nullptr;
        size_t size = 0;
    };
    virtual bool gather_adjoint(GatherAdjoint &) const { return false; }
    virtual ~Special() = default;
};

template <typename Value> struct Tape<Value>::Edge {
    Index source = 0;
    Value weight;
    std::unique_ptr<Special> special;

    Edge() = default;
    Edge(Index source, const Value &weight) : source(source), weight(weight) { }
    Edge(Index source, Special *special) : source(source), special(special) { }
    Edge(Edge &&) = default;
    Edge &operator=(Edge &&) = default;
    bool is_special() const { return special != nullptr; }
};

template <typename Value> struct Tape<Value>::Node {
    std::string label;
    Value grad;
    Value grad_weight;                // non-empty: the gradient is the pending product safe_mul(grad_weight, grad)
    std::vector<Edge> edges;          // incoming: this node = f(edge.source ...)
    std::vector<Index> consumers;     // nodes that have an edge from this node
    uint32_t ref_ext = 0, ref_int = 0;
    uint32_t size = 0;

    Node() = default;
    Node(size_t size, const char *label) : label(label ? label : ""), size((uint32_t) size) { }
    Node(Node &&) = default;
    Node &operator=(Node &&) = default;

    Edge *find_edge(Index source) {
        for (Edge &e : edges)
            if (e.source == source) return &e;
        return nullptr;
    }

    Edge take_edge(Index source) {
        for (auto it = edges.begin(); it != edges.end(); ++it) {
            if (it->source == source) {
                Edge e(std::move(*it));
                edges.erase(it);
                return e;
            }
        }
        throw std::runtime_error("Tape: internal error -- edge not found");
    }

    uint32_t elimination_cost() const { return (uint32_t) (edges.size() * consumers.size()); }
};

template <typename Value> struct Tape<Value>::Detail {
    Index next_index = 1, sweep_base = 1;
    std::unordered_map<Index, Node> nodes;
    std::vector<std::string> prefix;
    std::vector<Index> scheduled;               // sorted ascending, unique, once finalized
    std::unordered_map<Index, bool> visited;
    Index *operand_index = nullptr;             // scatter/gather operand (array_struct.h protocol)
    size_t operand_size = 0;
    bool operand_permute = false;
    uint32_t log_level = 0;
    bool simplification_enabled = true, simplified = true;

    Node &node(Index i) {
        auto it = nodes.find(i);
        if (it == nodes.end())
            throw std::runtime_error("autodiff: unknown variable index " + std::to_string(i));
        return it->second;
    }

    /// Collect everything reachable from `root` along incoming (backward) or outgoing (forward) edges
    void schedule(Index root, bool backward, bool clear_grad) {
        std::vector<Index> stack{ root };
        while (!stack.empty()) {
            Index k = stack.back();
            stack.pop_back();
            if (visited.count(k)) continue;
            visited[k] = true;
            scheduled.push_back(k);
            Node &n = node(k);
            if (clear_grad) { n.grad = Value(); n.grad_weight = Value(); }
            if (backward) {
                for (const Edge &e : n.edges) stack.push_back(e.source);
            } else {
                for (Index c : n.consumers) stack.push_back(c);
            }
        }
        std::sort(scheduled.begin(), scheduled.end());
    }

    void clear_schedule() {
        scheduled.clear();
        visited.clear();
    }

    static void accumulate(Value &dst, const Value &v) {
        if (dst.empty()) dst = v; else dst = dst + v;
    }

    // ---- deferred adjoints of gathers (backward sweep) ---------------------------------------------------------
    // The adjoint of `t = gather(source, offset, mask)` is `grad(source)[offset] += grad(t)`.  The sweep does not run
    // it on the spot: consecutive gather nodes that share their index array AND mask (e.g. a = gather(A, idx),
    // b = gather(B, idx)) are collected and handed to the backend as ONE multi-table scatter_add, which reads and
    // bins the indices once.  A gather node's gradient may also still be a pending edge product w * g (see
    // Node::grad_weight); the backend multiplies while it reads instead of materialising the product array.
    // Backends without scatter_add_multi_ execute the same adjoints one by one -- identical results.
    struct PendingScatter {
        Index source;
        Value grad, weight;
        Offset offset;
        Mask mask;
        size_t size;
    };
    std::vector<PendingScatter> pending;
    static constexpr size_t MaxPending = 4;

    static constexpr bool HasScatterAddMulti = detail::has_scatter_add_multi<Value, Offset>::value;

    template <typename T> static bool same_storage(const T &a, const T &b) {
        if constexpr (HasScatterAddMulti) return a.same_storage_(b);
        else return false;
    }

    /// Does `n` compute nothing but a non-permuting gather?
    static bool is_gather_node(const Node &n, typename Special::GatherAdjoint &ga) {
        return n.edges.size() == 1 && n.edges[0].is_special() && n.edges[0].special->gather_adjoint(ga);
    }

    bool pending_targets(Index i) const {
        for (const PendingScatter &p : pending)
            if (p.source == i) return true;
        return false;
    }

    void defer_gather_adjoint(Index source, const Value &grad, const Value &weight, const typename Special::GatherAdjoint &ga) {
        if (!pending.empty()) {
            const PendingScatter &first = pending[0];
            if (pending.size() == MaxPending || first.size != ga.size || pending_targets(source) ||
                !same_storage(first.offset, *ga.offset) || !same_storage(first.mask, *ga.mask))
                flush_pending();
        }
        pending.push_back(PendingScatter{ source, grad, weight, *ga.offset, *ga.mask, ga.size });
    }

    void flush_pending() {
        if (pending.empty()) return;
        std::vector<PendingScatter> work;
        work.swap(pending);
        Value *targets[MaxPending];
        const Value *values[MaxPending], *weights[MaxPending];
        bool any_weight = false;
        for (size_t c = 0; c < work.size(); ++c) {
            PendingScatter &p = work[c];
            materialize_grad(node(p.source));          // a gather of a gather: the table's gradient may be a pending product
            Value &grad_source = node(p.source).grad;
            if (grad_source.empty())
                grad_source = zero<Value>(p.size);
            else if (grad_source.size() == 1 && p.size != 1)
                set_slices(grad_source, p.size);           // pending broadcast contribution
            else if (grad_source.size() != p.size)
                throw std::runtime_error("Internal error in Gather::backward()!");
            targets[c] = &grad_source;
            values[c] = &p.grad;
            weights[c] = p.weight.empty() ? nullptr : &p.weight;
            any_weight = any_weight || weights[c];
        }
        if constexpr (HasScatterAddMulti) {
            if (work.size() > 1 || any_weight) {
                Value::scatter_add_multi_(work.size(), targets, values, weights, work[0].offset, work[0].mask);
                return;
            }
        }
        for (size_t c = 0; c < work.size(); ++c)
            scatter_add(*targets[c], weights[c] ? safe_mul(*weights[c], *values[c]) : *values[c], work[c].offset, work[c].mask);
    }

    /// Turn a pending edge product into an ordinary gradient array
    static void materialize_grad(Node &n) {
        if (!n.grad_weight.empty()) {
            n.grad = safe_mul(n.grad_weight, n.grad);
            n.grad_weight = Value();
        }
    }
};

template <typename Value> std::unique_ptr<Tape<Value>> Tape<Value>::s_tape;

template <typename Value> Tape<Value> *Tape<Value>::get() {
    if (!s_tape) s_tape = std::unique_ptr<Tape>(new Tape());
    return s_tape.get();
}

template <typename Value> Tape<Value>::Tape() : d(new Detail()) { }

template <typename Value> Tape<Value>::~Tape() {
    if (d->log_level >= 1 && !d->nodes.empty())
        std::cerr << "autodiff: " << d->nodes.size() << " variables were still live at shutdown." << std::endl;
    // Release arrays while the backend library is still loaded
    d->nodes.clear();
    delete d;
}

template <typename Value> void Tape<Value>::set_log_level(uint32_t level) { d->log_level = level; }
template <typename Value> uint32_t Tape<Value>::log_level() const { return d->log_level; }
template <typename Value> void Tape<Value>::set_graph_simplification(bool v) { d->simplification_enabled = v; }
template <typename Value> size_t Tape<Value>::node_count() const { return d->nodes.size(); }

// ---------------------------------------------------------------------------------------------
//  Recording
// ---------------------------------------------------------------------------------------------
template <typename Value> auto Tape<Value>::append_node(size_t size, const char *label) -> Index {
    Index idx = d->next_index++;
    Node &n = d->nodes.emplace(idx, Node(size, label)).first->second;
    for (auto it = d->prefix.rbegin(); it != d->prefix.rend(); ++it)
        n.label = *it + '/' + n.label;
    if (d->log_level >= 3)
        std::cerr << "autodiff: append_node(\"" << n.label << "\", size=" << size << ") -> " << idx << std::endl;
    inc_ref_ext(idx);
    d->simplified = false;
    return idx;
}

template <typename Value> auto Tape<Value>::append_leaf(size_t size) -> Index {
    // The reference zero-fills a size-N gradient here (autodiff.cpp:332-338) that the next sweep
    // discards again (dfs clears it, 177-182).  An immediate zero has the same observable value
    // without the N*4-byte write.
    Index idx = append_node(size, "'unnamed'");
    d->node(idx).grad = zero<Value>(1);
    return idx;
}

template <typename Value> auto Tape<Value>::append(const char *label, size_t size, Index i1, const Value &w1) -> Index {
    if (i1 == 0) return 0;
    Index idx = append_node(size, label);
    append_edge(i1, idx, w1);
    return idx;
}

template <typename Value>
auto Tape<Value>::append(const char *label, size_t size, Index i1, Index i2, const Value &w1, const Value &w2) -> Index {
    if (i1 == 0 && i2 == 0) return 0;
    Index idx = append_node(size, label);
    append_edge(i1, idx, w1);
    append_edge(i2, idx, w2);
    return idx;
}

template <typename Value>
auto Tape<Value>::append(const char *label, size_t size, Index i1, Index i2, Index i3, const Value &w1,
                         const Value &w2, const Value &w3) -> Index {
    if (i1 == 0 && i2 == 0 && i3 == 0) return 0;
    Index idx = append_node(size, label);
    append_edge(i1, idx, w1);
    append_edge(i2, idx, w2);
    append_edge(i3, idx, w3);
    return idx;
}

template <typename Value> void Tape<Value>::append_edge(Index source, Index target, const Value &weight) {
    if (source == 0) return;
    Node &t = d->node(target);
    if (Edge *e = t.find_edge(source)) {
        e->weight = e->weight + weight;       // x*x style duplicates merge (autodiff.cpp:624-632)
    } else {
        t.edges.emplace_back(source, weight);
        inc_ref_int(source, target);
    }
}

template <typename Value>
void Tape<Value>::append_edge_prod(Index source, Index target, const Value &w1, const Value &w2) {
    if (source == 0) return;
    Node &t = d->node(target);
    if (Edge *e = t.find_edge(source)) {
        e->weight = safe_fmadd(w1, w2, e->weight);
    } else {
        t.edges.emplace_back(source, safe_mul(w1, w2));
        inc_ref_int(source, target);
    }
}

template <typename Value> void Tape<Value>::set_label(Index idx, const char *label) {
    if (idx == 0) return;
    d->node(idx).label = "'" + std::string(label) + "'";
}

template <typename Value> void Tape<Value>::push_prefix(const char *value) { d->prefix.push_back(value); }
template <typename Value> void Tape<Value>::pop_prefix() {
    if (d->prefix.empty()) throw std::runtime_error("pop_prefix(): prefix list is already empty!");
    d->prefix.pop_back();
}

template <typename Value> void Tape<Value>::set_scatter_gather_operand(Index *index, size_t size, bool permute) {
    if (index != nullptr && d->operand_index != nullptr)
        throw std::runtime_error("set_scatter_gather_operand(): attempted to override an existing operand!");
    d->operand_index = index;
    d->operand_size = size;
    d->operand_permute = permute;
}

// ---------------------------------------------------------------------------------------------
//  Special edges: gather, scatter / scatter_add, reverse, prefix sum (autodiff.cpp:354-608)
// ---------------------------------------------------------------------------------------------
template <typename Value> auto Tape<Value>::append_gather(const Offset &offset, const Mask &mask) -> Index {
    if (d->operand_index == nullptr || *d->operand_index == 0) return 0;
    Index source = *d->operand_index;

    struct Gather : Special {
        Offset offset;
        Mask mask;
        size_t size;
        bool permute;

        // adjoint of a gather: scatter_add into a zeroed buffer of the source's size
        void backward(Detail *detail, Index target, const Edge &edge) const override {
            const Value &grad_target = detail->node(target).grad;
            Value &grad_source = detail->node(edge.source).grad;
            if (grad_source.empty())
                grad_source = zero<Value>(size);
            else if (grad_source.size() == 1 && size != 1)
                set_slices(grad_source, size);           // pending broadcast contribution
            else if (grad_source.size() != size)
                throw std::runtime_error("Internal error in Gather::backward()!");
            if (permute) scatter(grad_source, grad_target, offset, mask);
            else         scatter_add(grad_source, grad_target, offset, mask);
        }

        void forward(Detail *detail, Index target, const Edge &edge) const override {
            const Value &grad_source = detail->node(edge.source).grad;
            Value &grad_target = detail->node(target).grad;
            if (grad_source.size() != size && grad_source.size() != 1)
                throw std::runtime_error("Internal error in Gather::forward()!");
            Detail::accumulate(grad_target, gather<Value>(grad_source, offset, mask));
        }

        bool gather_adjoint(typename Special::GatherAdjoint &out) const override {
            if (permute) return false;
            out.offset = &offset;
            out.mask = &mask;
            out.size = size;
            return true;
        }
    };

    Gather *g = new Gather();
    g->offset = offset;
    g->mask = mask;
    g->size = d->operand_size;
    g->permute = d->operand_permute;

    Index target = append_node(std::max(slices(offset), slices(mask)), "gather");
    d->node(target).edges.emplace_back(source, g);
    inc_ref_int(source, target);
    return target;
}

template <typename Value>
void Tape<Value>::append_scatter(Index source, const Offset &offset, const Mask &mask, bool is_add) {
    if (d->operand_index == nullptr || source == 0) return;
    bool saved = d->simplification_enabled;
    d->simplification_enabled = false;
    Index target_orig = *d->operand_index;

    struct Scatter : Special {
        Offset offset;
        Mask mask;
        size_t size;
        bool is_add;

        void forward(Detail *detail, Index target, const Edge &edge) const override {
            const Value &grad_source = detail->node(edge.source).grad;
            Value &grad_target = detail->node(target).grad;
            if (grad_target.empty()) grad_target = zero<Value>(size);
            if (grad_target.size() == 1) set_slices(grad_target, size);
            if (grad_target.size() != size) throw std::runtime_error("Internal error in Scatter::forward()!");
            if (is_add) enoki::scatter_add(grad_target, grad_source, offset, mask);
            else        enoki::scatter(grad_target, grad_source, offset, mask);
        }

        // adjoint of a scatter: gather from the target's gradient
        void backward(Detail *detail, Index target, const Edge &edge) const override {
            Node &source = detail->node(edge.source);
            const Value &grad_target = detail->node(target).grad;
            if (grad_target.size() != size && grad_target.size() != 1)
                throw std::runtime_error("Internal error in Scatter::backward()!");
            Value result = gather<Value>(grad_target, offset, mask);
            if (source.size == 1 && result.size() != 1) result = hsum(result);
            Detail::accumulate(source.grad, result);
        }
    };

    Scatter *s = new Scatter();
    s->offset = offset;
    s->mask = mask;
    s->size = d->operand_size;
    s->is_add = is_add;

    Index target_new = append_node(d->operand_size, is_add ? "scatter_add" : "scatter");
    d->node(target_new).edges.emplace_back(source, s);
    inc_ref_int(source, target_new);

    if (target_orig != 0) {
        // combine with what the target held before: overwritten entries lose their old gradient
        Index scatter_node = target_new;
        Value weight = scalar_t<Value>(1);
        if (!is_add && !d->operand_permute) {
            weight = full<Value>(scalar_t<Value>(1), d->operand_size);
            scatter(weight, Value(scalar_t<Value>(0)), offset, mask);
        }
        target_new = append("scatter_combine", d->operand_size, target_new, target_orig, Value(scalar_t<Value>(1)), weight);
        dec_ref_ext(scatter_node);
        dec_ref_ext(target_orig);
    }
    *d->operand_index = target_new;
    d->simplification_enabled = saved;
}

template <typename Value> auto Tape<Value>::append_reverse(Index source) -> Index {
    if (source == 0) return 0;
    struct Reverse : Special {
        void forward(Detail *detail, Index target, const Edge &edge) const override {
            Detail::accumulate(detail->node(target).grad, reverse(detail->node(edge.source).grad));
        }
        void backward(Detail *detail, Index target, const Edge &edge) const override {
            Detail::accumulate(detail->node(edge.source).grad, reverse(detail->node(target).grad));
        }
    };
    Index target = append_node(d->node(source).size, "reverse");
    d->node(target).edges.emplace_back(source, new Reverse());
    inc_ref_int(source, target);
    return target;
}

template <typename Value>
auto Tape<Value>::append_custom(Index source, size_t size, const char *label, std::function<Value(const Value &)> backward) -> Index {
    if (source == 0) return 0;
    struct Custom : Special {
        std::function<Value(const Value &)> bwd;
        void forward(Detail *, Index, const Edge &) const override {
            throw std::runtime_error("autodiff: forward-mode traversal through a custom node is not available");
        }
        void backward(Detail *detail, Index target, const Edge &edge) const override {
            Detail::accumulate(detail->node(edge.source).grad, bwd(detail->node(target).grad));
        }
    };
    Custom *c = new Custom();
    c->bwd = std::move(backward);
    Index target = append_node(size, label);
    d->node(target).edges.emplace_back(source, c);
    inc_ref_int(source, target);
    return target;
}

template <typename Value> auto Tape<Value>::append_psum(Index source) -> Index {
    if (source == 0) return 0;
    struct PrefixSum : Special {
        void forward(Detail *detail, Index target, const Edge &edge) const override {
            Detail::accumulate(detail->node(target).grad, psum(detail->node(edge.source).grad));
        }
        void backward(Detail *detail, Index target, const Edge &edge) const override {
            Node &t = detail->node(target);
            Value g = t.grad;
            if (g.size() == 1 && t.size != 1) set_slices(g, t.size);
            Detail::accumulate(detail->node(edge.source).grad, reverse(psum(reverse(g))));
        }
    };
    Index target = append_node(d->node(source).size, "psum");
    d->node(target).edges.emplace_back(source, new PrefixSum());
    inc_ref_int(source, target);
    return target;
}

// ---------------------------------------------------------------------------------------------
//  Reference counting (autodiff.cpp:681-774)
// ---------------------------------------------------------------------------------------------
template <typename Value> void Tape<Value>::inc_ref_int(Index index, Index from) {
    Node &n = d->node(index);
    if (std::find(n.consumers.begin(), n.consumers.end(), from) != n.consumers.end())
        throw std::runtime_error("inc_ref_int(): internal error -- edge already exists!");
    n.consumers.push_back(from);
    n.ref_int++;
}

template <typename Value> void Tape<Value>::dec_ref_int(Index index, Index from) {
    if (index == 0) return;
    Node &n = d->node(index);
    if (n.ref_int == 0)
        throw std::runtime_error("autodiff: dec_ref_int(): node " + std::to_string(index) + " has no internal references!");
    --n.ref_int;
    auto it = std::find(n.consumers.begin(), n.consumers.end(), from);
    if (it == n.consumers.end())
        throw std::runtime_error("dec_ref_int(): internal error -- edge not found!");
    n.consumers.erase(it);
    if (n.ref_int == 0 && n.ref_ext == 0) free_node(index);
}

template <typename Value> void Tape<Value>::inc_ref_ext(Index index) {
    if (index == 0) return;
    d->node(index).ref_ext++;
}

template <typename Value> void Tape<Value>::dec_ref_ext(Index index) {
    if (index == 0) return;
    Node &n = d->node(index);
    if (n.ref_ext == 0)
        throw std::runtime_error("autodiff: dec_ref_ext(): node " + std::to_string(index) + " has no external references!");
    --n.ref_ext;
    if (n.ref_int == 0 && n.ref_ext == 0) free_node(index);
}

template <typename Value> void Tape<Value>::free_node(Index index) {
    auto it = d->nodes.find(index);
    if (it == d->nodes.end())
        throw std::runtime_error("autodiff: free_node(): unknown index " + std::to_string(index));
    // detach first: dec_ref_int may recursively free ancestors and rehash the map
    std::vector<Edge> edges = std::move(it->second.edges);
    d->nodes.erase(it);
    for (const Edge &e : edges) dec_ref_int(e.source, index);
}

// ---------------------------------------------------------------------------------------------
//  Sweeps
// ---------------------------------------------------------------------------------------------
template <typename Value> const Value &Tape<Value>::gradient(Index index) {
    if (index == 0)
        throw std::runtime_error("No gradient was computed for this variable! (a call to requires_gradient() is necessary.)");
    return d->node(index).grad;
}

template <typename Value> void Tape<Value>::set_gradient(Index index, const Value &value, bool backward) {
    if (index == 0)
        throw std::runtime_error("set_gradient(): no gradients are associated with this variable (a prior call to "
                                 "requires_gradient() is required.)");
    d->schedule(index, backward, true);
    d->node(index).grad = value;
}

template <typename Value> void Tape<Value>::backward(Index index, bool free_graph) {
    bool saved = d->simplification_enabled;
    d->simplification_enabled = false;
    set_gradient(index, Value(scalar_t<Value>(1)), true);
    backward(free_graph);
    d->simplification_enabled = saved;
}

template <typename Value> void Tape<Value>::forward(Index index, bool free_graph) {
    bool saved = d->simplification_enabled;
    d->simplification_enabled = false;
    set_gradient(index, Value(scalar_t<Value>(1)), false);
    forward(free_graph);
    d->simplification_enabled = saved;
}

namespace detail {
    /// Brackets a sweep for backends that want to know (HIPArray: scatters inside a sweep always copy on write)
    template <typename Value, typename = void> struct SweepScope { };
    template <typename Value> struct SweepScope<Value, std::void_t<decltype(Value::sweep_scope_(true))>> {
        SweepScope() { Value::sweep_scope_(true); }
        ~SweepScope() { Value::sweep_scope_(false); }
    };
}

template <typename Value> void Tape<Value>::backward(bool free_graph) {
    detail::SweepScope<Value> sweep_scope;
    std::vector<Index> order = d->scheduled;
    d->clear_schedule();
    d->pending.clear();             // (left over only if an earlier sweep threw)

    if (free_graph)
        for (Index i : order) inc_ref_ext(i);

    for (auto it = order.rbegin(); it != order.rend(); ++it) {
        Index target_idx = *it;
        Node &target = d->node(target_idx);

        // gather nodes queue their adjoint (see Detail::PendingScatter); anything else first runs what is queued
        typename Special::GatherAdjoint gather_adjoint;
        const bool is_gather = Detail::is_gather_node(target, gather_adjoint);
        if (!is_gather || d->pending_targets(target_idx)) d->flush_pending();
        if (!is_gather) Detail::materialize_grad(target);

        if (target.grad.size() != target.size) {
            if (target.grad.size() > 1)
                throw std::runtime_error("backward(): gradient sizes don't match: expected " +
                                         std::to_string(target.size) + ", got " + std::to_string(target.grad.size()));
            // a size-1 gradient of a vector node is a broadcast.  Interior nodes hand it to kernels as
            // a broadcast operand; only gradients that stay visible (leaves) are materialised.
            if (target.edges.empty() && target.grad.size() == 1) set_slices(target.grad, target.size);
        }
        const bool has_grad = !target.grad.empty();   // false: nothing reached this node

        for (Edge &edge : target.edges) {
            Node &source = d->node(edge.source);
            if (!has_grad) {
                /* nothing to propagate */
            } else if (!edge.is_special()) {
                if (source.size == 1 && (edge.weight.size() != 1 || target.grad.size() != 1 || target.size != 1)) {
                    // vector target feeding a scalar source: the contribution is a horizontal sum over the
                    // target's entries.  The reference materialises size-1 gradients to the node size first
                    // (autodiff.cpp:851-853); here they stay broadcast, so when weight AND gradient are both
                    // broadcasts the sum of target.size equal terms is formed as a product.
                    Value contribution = hsum_safe_mul(edge.weight, target.grad);
                    if (edge.weight.size() == 1 && target.grad.size() == 1 && target.size != 1)
                        contribution = contribution * Value(scalar_t<Value>(target.size));
                    Detail::accumulate(source.grad, contribution);
                } else if (source.grad.empty()) {
                    // unit weight (add/sub/fmadd addend/...) or unit gradient (the seed of backward(), passed on
                    // through hsum): w * 1 and 1 * g -- share the other operand's buffer, no kernel
                    typename Special::GatherAdjoint ga;
                    if (detail::is_unit_weight(edge.weight)) {
                        source.grad = target.grad;
                    } else if (detail::is_unit_weight(target.grad)) {
                        source.grad = edge.weight;
                    } else if (Detail::HasScatterAddMulti && source.size > 1 && Detail::is_gather_node(source, ga) &&
                               std::max(edge.weight.size(), target.grad.size()) == source.size) {
                        // the product is only ever read by the gather's adjoint: leave it pending, the scatter_add
                        // multiplies while it reads (a second contribution materialises it, see below)
                        source.grad = target.grad;
                        source.grad_weight = edge.weight;
                    } else {
                        source.grad = safe_mul(edge.weight, target.grad);
                    }
                } else {
                    Detail::materialize_grad(source);
                    source.grad = safe_fmadd(edge.weight, target.grad, source.grad);
                }
            } else if (is_gather) {
                d->defer_gather_adjoint(edge.source, target.grad, target.grad_weight, gather_adjoint);
            } else {
                // specials accumulate straight into source.grad: a pending product w * g must become an array first
                // (otherwise the result would be w * (g + contribution))
                Detail::materialize_grad(source);
                edge.special->backward(d, target_idx, edge);
            }
            if (free_graph) {
                dec_ref_int(edge.source, target_idx);
                edge.source = 0;
            }
        }

        if (free_graph) {
            Node &t = d->node(target_idx);      // re-lookup: the map may have rehashed
            if (!t.edges.empty()) {
                t.edges.clear();
                t.grad = Value();
                t.grad_weight = Value();
            }
            dec_ref_ext(target_idx);
        } else if (target.ref_int > 0 && !target.edges.empty()) {
            target.grad = Value();
            target.grad_weight = Value();
        } else {
            Detail::materialize_grad(target);   // the gradient stays visible
        }
    }
    d->flush_pending();

    if (d->log_level >= 1)
        std::cerr << "autodiff: backward(): processed " << order.size() << "/" << (d->next_index - d->sweep_base)
                  << " nodes." << std::endl;
    if (free_graph) d->sweep_base = d->next_index;
}

template <typename Value> void Tape<Value>::forward(bool free_graph) {
    detail::SweepScope<Value> sweep_scope;
    std::vector<Index> order = d->scheduled;
    d->clear_schedule();

    if (free_graph)
        for (Index i : order) inc_ref_ext(i);

    for (Index source_idx : order) {
        Node &source = d->node(source_idx);
        if (source.size == 1 && source.grad.size() > 1) source.grad = hsum(source.grad);

        std::vector<Index> consumers = source.consumers;
        for (Index target_idx : consumers) {
            Node &target = d->node(target_idx);
            Edge *edge = target.find_edge(source_idx);
            if (!edge) throw std::runtime_error("forward(): invalid graph structure!");
            Node &src = d->node(source_idx);
            if (!src.grad.empty()) {
                if (!edge->is_special()) {
                    if (target.size == 1 && (edge->weight.size() != 1 || src.grad.size() != 1 || src.size != 1)) {
                        Value contribution = hsum_safe_mul(edge->weight, src.grad);
                        if (edge->weight.size() == 1 && src.grad.size() == 1 && src.size != 1)   // see backward()
                            contribution = contribution * Value(scalar_t<Value>(src.size));
                        Detail::accumulate(target.grad, contribution);
                    } else if (target.grad.empty()) {
                        target.grad = detail::is_unit_weight(edge->weight) ? src.grad
                                    : detail::is_unit_weight(src.grad)     ? edge->weight
                                                                           : safe_mul(edge->weight, src.grad);
                    } else {
                        target.grad = safe_fmadd(edge->weight, src.grad, target.grad);
                    }
                } else {
                    edge->special->forward(d, target_idx, *edge);
                }
            }
        }

        Node &s = d->node(source_idx);
        if (s.size != 1 && s.grad.size() == 1 && s.consumers.empty()) set_slices(s.grad, s.size);
        if (s.ref_int > 0) s.grad = Value();
        if (free_graph) {
            for (Index target_idx : consumers) {
                d->node(target_idx).take_edge(source_idx);
                dec_ref_int(source_idx, target_idx);
            }
            dec_ref_ext(source_idx);
        }
    }

    if (d->log_level >= 1)
        std::cerr << "autodiff: forward(): processed " << order.size() << "/" << (d->next_index - d->sweep_base)
                  << " nodes." << std::endl;
    if (free_graph) d->sweep_base = d->next_index;
}

// ---------------------------------------------------------------------------------------------
//  Graph simplification by greedy vertex elimination (autodiff.cpp:990-1074)
// ---------------------------------------------------------------------------------------------
template <typename Value> void Tape<Value>::simplify_graph() {
    if (d->simplified) return;
    bool saved = d->simplification_enabled;
    d->simplification_enabled = false;

    std::set<std::pair<uint32_t, Index>> todo;    // (cost, node), cheapest first
    for (const auto &kv : d->nodes) todo.emplace(kv.second.elimination_cost(), kv.first);
    size_t cost = 0;

    while (!todo.empty()) {
        auto [score, index] = *todo.begin();
        todo.erase(todo.begin());
        Node &node = d->node(index);
        if (node.edges.empty() || node.consumers.empty()) continue;      // leaves and outputs stay
        if (score > ENOKI_AUTODIFF_MAX_SIMPLIFICATION_COST) break;

        std::vector<std::pair<uint32_t, Index>> neighbours;
        bool skip = false;
        for (Index k : node.consumers) {
            Node &c = d->node(k);
            if (c.find_edge(index)->is_special()) skip = true;
            neighbours.emplace_back(c.elimination_cost(), k);
        }
        for (const Edge &e : node.edges) {
            const Node &p = d->node(e.source);
            neighbours.emplace_back(p.elimination_cost(), e.source);
            if ((node.size == 1 && p.size != node.size) || e.is_special()) skip = true;
        }
        if (skip) continue;

        std::vector<Index> consumers = node.consumers;
        for (Index other : consumers) {
            Edge outgoing = d->node(other).take_edge(index);
            for (const Edge &incoming : d->node(index).edges) {
                append_edge_prod(incoming.source, other, outgoing.weight, incoming.weight);
                cost++;
            }
            dec_ref_int(index, other);          // frees the node after its last consumer
        }

        for (auto [old_score, id] : neighbours) {
            auto it = todo.find({ old_score, id });
            if (it == todo.end() || d->nodes.find(id) == d->nodes.end()) continue;
            uint32_t new_score = d->node(id).elimination_cost();
            if (new_score != old_score) {
                todo.erase(it);
                todo.emplace(new_score, id);
            }
        }
    }

    if (d->log_level >= 2)
        std::cerr << "autodiff: simplify_graph(): done. (cost = " << cost << ")" << std::endl;
    d->simplified = true;
    d->simplification_enabled = saved;
}

// ---------------------------------------------------------------------------------------------
//  Introspection
// ---------------------------------------------------------------------------------------------
template <typename Value> std::string Tape<Value>::graphviz(const std::vector<Index> &roots) {
    std::ostringstream oss;
    oss << "digraph {\n  rankdir=BT;\n  node [shape=record fontname=Consolas];\n";
    for (Index r : roots) d->schedule(r, true, false);
    std::vector<Index> order = d->scheduled;
    d->clear_schedule();
    for (Index i : order) {
        const Node &n = d->node(i);
        oss << "  " << i << " [label=\"" << n.label << (n.size == 1 ? " [s]" : "") << "\\n#" << i << " [E/I: "
            << n.ref_ext << "/" << n.ref_int << "]\"";
        if (!n.label.empty() && n.label[0] == '\'') oss << " fillcolor=salmon style=filled";
        oss << "];\n";
    }
    for (Index i : order) {
        for (const Edge &e : d->node(i).edges) {
            oss << "  " << i << " -> " << e.source << ";\n";
            if (e.is_special()) oss << "  " << i << " [shape=doubleoctagon];\n";
        }
    }
    for (Index r : roots) oss << "  " << r << " [fillcolor=cornflowerblue style=filled];\n";
    oss << "}";
    return oss.str();
}

template <typename Value> std::string Tape<Value>::whos() const {
    std::vector<Index> ids;
    for (const auto &kv : d->nodes) ids.push_back(kv.first);
    std::sort(ids.begin(), ids.end());
    std::ostringstream oss;
    oss << "\n  ID      E/I Refs   Size        Label\n  ====================================\n";
    for (Index id : ids) {
        const Node &n = d->node(id);
        oss << "  " << std::left << std::setw(7) << id << " " << std::setw(10)
            << (std::to_string(n.ref_ext) + " / " + std::to_string(n.ref_int)) << " " << std::setw(12) << n.size
            << n.label << "\n";
    }
    oss << "  ====================================\n\n";
    return oss.str();
}

} // namespace enoki
