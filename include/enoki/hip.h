/*
    enoki/hip.h -- HIPArray<Value>: a 1-D array resident in MI355X memory

    Plays the role of the reference's CUDAArray<Value> (include/enoki/cuda.h:205-954) and keeps its
    member concept (`add_`, `fmadd_`, `select_`, `gather_<Stride>`, `hsum_`, `map`, `copy`, ...), but
    NOT its mechanism: there is no trace, no JIT and no variable table.  A HIPArray owns a
    reference-counted device buffer; every operation launches one pre-compiled HIP kernel from
    libenoki-hip.so (include/enoki_hip.h) on the library stream and returns a new array.

      * copies share the buffer (refcount), like cuda.h:224-226;
      * arrays of size 1 created from host values are *immediates*: they live in the handle and
        are passed to kernels as arguments, so `a * 2.f` never materialises a broadcast;
      * arrays of size 1 produced on the device (hsum, ...) stay on the device -- nothing here
        synchronises except coeff()/operator[], all_/any_/count_ and the *_to_host copies;
      * size mismatches throw std::runtime_error("arrays of incompatible size"), jit.cu:776-782.
*/
#pragma once

#include <enoki/array.h>
#include <enoki_hip.h>

#include <cmath>
#include <cstdlib>
#include <initializer_list>
#include <vector>

namespace enoki {

namespace detail {
    [[noreturn]] inline void hip_raise(const char *what) {
        throw std::runtime_error(std::string(what) + ": " + ek_hip_last_error());
    }
    inline void hip_check(int rc, const char *what) {
        if (rc != EK_OK) hip_raise(what);
    }
    /// ENOKI_HIP_LOG >= 2 (hip_set_log_level): one line whenever an expression that could have run in bucket order does not --
    /// which shape was expected and what was found instead (`explain()` of an array says what state it is in)
    inline void hip_note_element_order(const char *what, const char *why) {
        if (ek_hip_log_level() >= 2) fprintf(stderr, "enoki-hip: [bucket order] %s: %s\n", what, why);
    }

    /// Reference-counted device allocation
    ///
    /// A buffer may be a *deferred gather*: `table[index]` under `mask` that has not been executed yet (ptr == nullptr).
    /// The reference never writes a gather that feeds an arithmetic op to memory -- its JIT fuses the load into the
    /// consumer (jit.cu:1066-1217) -- and neither does HIPArray: the first add/sub/mul/fma that consumes a deferred gather
    /// reads the table in place (ek_hip_map_gathered); any other access (data(), operand(), a second consumer, ...) runs
    /// the plain gather kernel first.  Arrays stay immutable values: the deferred node holds references on its table,
    /// index and mask buffers, and a buffer with deferred readers forces them before it hands out a mutable pointer.
    struct HIPBuffer {
        void *ptr = nullptr;
        size_t size = 0;          // elements
        uint32_t ref_count = 1;
        bool owned = true;        // false for map()ped memory that the caller keeps

        ///
        /// The same mechanism carries *deferred unary maps* (kind 1): `op(source)` for the ops a streaming consumer can
        /// apply while it loads (neg, abs, sqrt, rcp, rsqrt, sin, cos, exp, log).  Horizontal reductions and the value
        /// streams of scatter_add consume them in place (ek_hip_reduce_map, ek_hip_scatter_add_multi_map):
        /// `hsum(sin(u))` reads u once and writes nothing, and the cos(u) that `backward()` needs as the derivative is
        /// evaluated inside the adjoint scatter_add of the gathers that produced u.  Any other access runs the plain
        /// kernel -- for the two halves of a sincos() ONE kernel that fills both, as before.
        struct Deferred {
            HIPBuffer *table, *index, *mask;   // references held; mask == nullptr: every lane is active.  Map: table = source
            int type, index_type;              // map: index_type holds the unary op
            size_t elem_size;
            bool consumed;                     // a fused consumer has read it once: the next access materialises (gathers)
            int kind = 0;                      // 0: gather, 1: unary map, 2: fma of a gathered pair (below), 3: zeros (no sources),
                                               // 4: arithmetic over evaluated operands (below)
            HIPBuffer *partner = nullptr;      // map: the other half of an unevaluated sincos pair (not owning)
            // map: the node is  scale * op(source)  -- the product of an unevaluated map with a host scalar stays a map
            // (HIPArray::scaled_map_: the -sin(u) that d/du cos(u) records, the c * cos(u) that backward(c * y) sends down),
            // so that its consumers still see WHICH function of the source it is.  Bits of the element type.
            uint64_t scale_bits = 0;
            bool scaled = false;
            // kind 2:  u = op(table[index], arg0, table2[index])  with op of the fma family -- the parameter lookup
            // `fmadd(gather(A, idx), x, gather(B, idx))`.  Left unevaluated one step longer than its gathers: when the
            // consumer does not care about the element order (a horizontal reduction, possibly through a deferred unary map;
            // the adjoint scatter_add of the two gathers through the same index array) the chain runs BUCKET BY BUCKET out
            // of LDS-resident table slices (ek_hip_bucketed_*) and the partition is kept here for the backward sweep.  Every
            // other access evaluates u in element order with the kernels that consume a gather in place (same bits).
            HIPBuffer *table2 = nullptr, *arg0 = nullptr;      // references held
            int op = 0;
            ek_hip_bucketed *bucketed = nullptr;               // the partition of (index, arg0) by table bucket, once built
            // kind 4:  op(operand 0, operand 1[, operand 2])  over EVALUATED arrays / host scalars that has not run yet -- the fma
            // family, a product, and the product-then-sum forms EK_MULADD / EK_MULSUB / EK_NMULADD that `a * x + b` written with
            // operators becomes.  Operand k is the buffer in table / arg0 / table2 (references held) or the immediate imm[k].
            // The reference fuses every vertical op between two evaluation points into one kernel (jit.cu:1066-1217,
            // :1418-1508); here a horizontal reduction that finds such a node under (up to three) unevaluated unary maps reads
            // the operands once and writes nothing (ek_hip_reduce_chain; `hsum(sin(exp(fmadd(a, x, b))))`, BASELINE configs[1]:
            // 12 B/elt instead of 28), and any other access runs the plain kernel -- same bits either way.
            // kind 2 WITHOUT an addend (table2 == nullptr, op == EK_MULADD):  table[index] * arg0  -- the product `gather(A, idx) * x`.
            // A sum that adds a gather through the same index array turns it into the full kind-2 node (op EK_MULADD / EK_MULSUB /
            // EK_NMULADD: `gather(A, idx) * x + gather(B, idx)` written with operators, a product and a sum with a rounding each); a
            // horizontal reduction (through one fusable unary op) and the adjoint scatter_add of the ONE gather take it in bucket order
            // as it is; any other access runs the kernel that consumes the gather in place, as the eager product did.
            int arity = 0;
            uint64_t imm[3] = { 0, 0, 0 };
            bool is_imm[3] = { false, false, false };
            HIPBuffer *operand_buf(int k) const { return k == 0 ? table : k == 1 ? arg0 : table2; }
        };
        Deferred *deferred = nullptr;
        std::vector<HIPBuffer *> readers;      // deferred nodes whose table / source is THIS buffer (not owning)
        HIPBuffer *pending_prev = nullptr, *pending_next = nullptr;   // list of all unevaluated buffers (see pending_head())

        /// Every buffer that is still unevaluated, newest first.  hip_graph_begin() evaluates them all: a node that predates a
        /// capture must not receive its storage from the graph's private pool (it would dangle once the graph is destroyed).
        /// (The head lives in libenoki-hip.so, ek_hip_binding_slot(): this header is compiled into several shared objects --
        /// the two python modules, the tape library, user code -- that hand buffers to each other.)
        struct Shared {
            HIPBuffer *pending = nullptr;
            bool defer = true;
            bool scatter_alias = false;      // hip_set_scatter_aliasing(): user-level scatters write in place through shared handles
            int sweeps = 0;                  // > 0 while a Tape sweep runs (its buffers are shared as VALUES: always copy on write)
        };
        static Shared &shared() {
            void **slot = ek_hip_binding_slot();
            if (!*slot) {
                Shared *s = new Shared();             // once per process, never freed
                const char *e = getenv("ENOKI_HIP_DEFER"), *g = getenv("ENOKI_HIP_DEFER_GATHER");
                s->defer = !((e && e[0] == '0') || (g && g[0] == '0'));
                const char *al = getenv("ENOKI_HIP_SCATTER_ALIASING");
                s->scatter_alias = al && al[0] == '1';
                *slot = s;
            }
            return *static_cast<Shared *>(*slot);
        }
        static HIPBuffer *&pending_head() { return shared().pending; }
        void pending_link() {
            pending_prev = nullptr;
            pending_next = pending_head();
            if (pending_next) pending_next->pending_prev = this;
            pending_head() = this;
        }
        void pending_unlink() {
            if (pending_prev) pending_prev->pending_next = pending_next;
            else if (pending_head() == this) pending_head() = pending_next;
            if (pending_next) pending_next->pending_prev = pending_prev;
            pending_prev = pending_next = nullptr;
        }
        static void force_all_pending() {
            while (pending_head()) pending_head()->force();
        }
        HIPBuffer *view_of = nullptr;          // a window into another buffer (HIPArray::view_): holds a reference on it
        // A 64-bit integer array that was narrowed to 32 bits keeps the result for as long as its contents cannot change (owning):
        // the gather, the tape's offset array (autodiff.h `Offset(index)`) and the adjoint scatter_add of ONE 64-bit index array then
        // all see the same 32-bit buffer -- the index array is narrowed once, and the bucket-ordered path recognises it.
        HIPBuffer *narrowed = nullptr;
        int narrowed_type = 0;
        HIPBuffer *narrowed_of = nullptr;      // this buffer IS some source's cache entry (not owning; cleared when the entry goes)
        void drop_narrowed() { if (narrowed) { HIPBuffer *n = narrowed; narrowed = nullptr; n->narrowed_of = nullptr; unref(n); } }
        /// A cache entry that is about to be written through one of its own handles (non-const data(), an in-place scatter) or
        /// exported stops being a cache entry: the source narrows again next time
        void leave_narrowed_cache() { if (narrowed_of) narrowed_of->drop_narrowed(); }
        void *host_mirror = nullptr;           // begin() / end(): read-only host copy, dropped when the buffer may change
        bool exported = false;                 // an external zero-copy view (torch, __cuda_array_interface__) may exist

        static void unref(HIPBuffer *b) {
            if (b && --b->ref_count == 0) delete b;
        }

        ek_gathered gathered() const {
            ek_gathered g;
            g.table = deferred->table->ptr;
            g.table_size = deferred->table->size;
            g.index = ek_operand{ deferred->index->ptr, 0, deferred->index->size };
            g.index_type = deferred->index_type;
            g.mask = deferred->mask ? ek_operand{ deferred->mask->ptr, 0, deferred->mask->size } : ek_operand{ nullptr, 1, 1 };
            return g;
        }

        /// Execute a deferred unary map; a sincos pair is evaluated by one kernel
        void force_map() {
            Deferred *d = deferred;
            const size_t bytes = (size ? size : 1) * d->elem_size;
            if (d->table->deferred && (d->table->deferred->kind == 1 || d->table->deferred->kind == 4) &&
                !(d->partner && d->partner->deferred)) {
                // maps over maps over an unevaluated arithmetic node: what only this chain wants is evaluated in ONE pass
                ek_chain ch;
                build_chain(ch);
                if (ch.n_maps > 1 || ch.arity > 1) {
                    void *p = nullptr;
                    hip_check(ek_hip_malloc(bytes, &p), "HIPArray (deferred chain)");
                    ek_operand factor{ nullptr, d->scale_bits, 1 };           // (a scaled node: one more rounding, inside the pass)
                    int rc = d->scaled ? ek_hip_map_chain_product(d->type, p, nullptr, &ch, &factor, EK_MUL, nullptr, size)
                                       : ek_hip_map_chain(d->type, p, &ch, size);
                    if (rc != EK_OK) {
                        ek_hip_free(p);
                        hip_raise("HIPArray (deferred chain)");
                    }
                    ptr = p;
                    drop_deferred();
                    return;
                }
            }
            if (d->table->deferred) d->table->force();          // a source that is an unevaluated fma of gathers runs first
            ek_operand src{ d->table->ptr, 0, d->table->size };
            HIPBuffer *other = d->partner && d->partner->deferred ? d->partner : nullptr;
            void *p = nullptr, *q = nullptr;
            hip_check(ek_hip_malloc(bytes, &p), "HIPArray (deferred map)");
            int rc;
            if (other) {
                rc = ek_hip_malloc(bytes, &q);
                if (rc == EK_OK) {
                    const bool is_sin = d->index_type == EK_SIN;
                    rc = ek_hip_sincos(d->type, is_sin ? p : q, is_sin ? q : p, &src, size);
                }
            } else {
                rc = ek_hip_unary(d->index_type, d->type, p, &src, size);
            }
            if (rc != EK_OK) {
                ek_hip_free(p);
                if (q) ek_hip_free(q);
                hip_raise("HIPArray (deferred map)");
            }
            // a scaled node: the product with its host scalar, in place (the same two roundings as op, then mul, run eagerly)
            auto scale_in_place = [&](HIPBuffer *b2, void *at) {
                if (!b2->deferred->scaled) return;
                ek_operand self{ at, 0, size }, factor{ nullptr, b2->deferred->scale_bits, 1 };
                if (ek_hip_binary(EK_MUL, d->type, at, &self, &factor, size) != EK_OK) {
                    ek_hip_free(p);
                    if (q) ek_hip_free(q);
                    hip_raise("HIPArray (deferred map)");
                }
            };
            scale_in_place(this, p);
            if (other) scale_in_place(other, q);
            ptr = p;
            if (other) {
                other->ptr = q;
                other->drop_deferred();
            }
            drop_deferred();
        }

        /// A deferred unary map whose first consumer is a product with an evaluated array `w` -- safe_mul(edge weight, gradient)
        /// of the backward sweep, with the gradient still the unevaluated cos(u) -- is evaluated TOGETHER with that product: one
        /// pass over the chain's operands writes both (ek_hip_map_chain_product); the caller owns `product` (size elements).
        /// Precondition: no unevaluated sincos partner (that pair has a kernel of its own).
        void force_map_product(const HIPBuffer *w, int op2, void *product) {
            Deferred *d = deferred;
            ek_chain ch;
            build_chain(ch);
            void *p = nullptr;
            hip_check(ek_hip_malloc((size ? size : 1) * d->elem_size, &p), "HIPArray (deferred map and its product)");
            ek_operand factor{ nullptr, d->scale_bits, 1 }, other{ w->ptr, 0, w->size };
            if (ek_hip_map_chain_product(d->type, p, product, &ch, d->scaled ? &factor : nullptr, op2, &other, size) != EK_OK) {
                ek_hip_free(p);
                hip_raise("HIPArray (deferred map and its product)");
            }
            ptr = p;
            drop_deferred();
        }

        /// Element-order evaluation of a deferred fma over a gathered pair: one kernel, gathers consumed in place
        void force_pair() {
            Deferred *d = deferred;
            hip_note_element_order("fma of two gathers through one index array",
                                   "evaluated in ELEMENT order -- its consumer is not a horizontal reduction (directly or through one "
                                   "fusable unary op) nor the adjoint scatter_add of its gathers, or a source is about to be written, or a "
                                   "step graph is being captured, or deterministic mode is on");
            if (!d->table2) { force_gathered_product(); return; }
            void *p = nullptr;
            hip_check(ek_hip_malloc((size ? size : 1) * d->elem_size, &p), "HIPArray (deferred fma of gathers)");
            ek_gathered ga, gc;
            ga.table = d->table->ptr;
            ga.table_size = d->table->size;
            ga.index = ek_operand{ d->index->ptr, 0, d->index->size };
            ga.index_type = d->index_type;
            ga.mask = d->mask ? ek_operand{ d->mask->ptr, 0, d->mask->size } : ek_operand{ nullptr, 1, 1 };
            gc = ga;
            gc.table = d->table2->ptr;
            gc.table_size = d->table2->size;
            ek_operand x{ d->arg0->ptr, 0, d->arg0->size };
            const ek_gathered *pg[3] = { &ga, nullptr, &gc };
            const ek_operand *po[3] = { nullptr, &x, nullptr };
            if (ek_hip_map_gathered(3, d->op, d->type, p, po, pg, size) != EK_OK) {
                ek_hip_free(p);
                hip_raise("HIPArray (deferred fma of gathers)");
            }
            ptr = p;
            drop_deferred();
        }

        /// The bucket partition of a kind-2 node (built on first use); nullptr when the library does not cover the shape
        ek_hip_bucketed *bucketed(unsigned hints = 0) {
            Deferred *d = deferred;
            if (!d->bucketed) {
                int rc = ek_hip_bucketed_pair_create_masked(d->type, d->index_type, d->op, d->table->ptr, d->table2 ? d->table2->ptr : nullptr, d->table->size,
                                                            d->arg0->ptr, d->index->ptr, d->mask ? (const uint8_t *) d->mask->ptr : nullptr,
                                                            size, hints, &d->bucketed);
                if (rc == EK_ERR_UNSUPPORTED) return nullptr;
                hip_check(rc, "HIPArray (bucket partition)");
            }
            return d->bucketed;
        }

        /// kind 4: run the deferred arithmetic op
        void force_arith() {
            Deferred *d = deferred;
            void *p = nullptr;
            hip_check(ek_hip_malloc((size ? size : 1) * d->elem_size, &p), "HIPArray (deferred arithmetic)");
            ek_operand o[3];
            for (int k = 0; k < d->arity; ++k) {
                HIPBuffer *b = d->operand_buf(k);
                o[k] = d->is_imm[k] ? ek_operand{ nullptr, d->imm[k], 1 } : ek_operand{ b->ptr, 0, b->size };
            }
            int rc = d->arity == 3 ? ek_hip_ternary(d->op, d->type, p, &o[0], &o[1], &o[2], size)
                                   : ek_hip_binary(d->op, d->type, p, &o[0], &o[1], size);
            if (rc != EK_OK) {
                ek_hip_free(p);
                hip_raise("HIPArray (deferred arithmetic)");
            }
            ptr = p;
            drop_deferred();
        }

        /// kind 2 without an addend: the product of a gather with an array, the gather consumed in place (element order)
        void force_gathered_product() {
            Deferred *d = deferred;
            void *p = nullptr;
            hip_check(ek_hip_malloc((size ? size : 1) * d->elem_size, &p), "HIPArray (deferred product of a gather)");
            ek_gathered g = gathered();
            ek_operand x{ d->arg0->ptr, 0, d->arg0->size };
            const ek_gathered *pg[3] = { &g, nullptr, nullptr };
            const ek_operand *po[3] = { nullptr, &x, nullptr };
            if (ek_hip_map_gathered(2, EK_MUL, d->type, p, po, pg, size) != EK_OK) {
                ek_hip_free(p);
                hip_raise("HIPArray (deferred product of a gather)");
            }
            ptr = p;
            drop_deferred();
        }

        /// Only the chain above holds / reads this unevaluated node: it can be recomputed inside the consumer instead of written
        bool absorbable_() const { return deferred && ref_count == 1 && readers.size() == 1; }
        /// An unevaluated arithmetic node that only `reader` and ONE more unevaluated map hold -- the sin(u) / cos(u) that
        /// differentiating sin records next to each other.  A reduction of the one reads u's operands (12 B/elt for an fma)
        /// instead of writing u for the other (16 + 4): no worse whatever the sibling does later, and better when the
        /// sibling in turn finds itself alone (hsum(sin(fmadd(a, x, b))) followed by backward(): cfg3a, 40 -> 32 B/elt).
        bool only_sibling_maps_(const HIPBuffer *reader) const {
            if (!deferred || deferred->kind != 4 || readers.size() != 2 || ref_count != 2) return false;
            for (const HIPBuffer *rd : readers)
                if (rd != reader && !(rd->deferred && rd->deferred->kind == 1 && rd->deferred->table == this)) return false;
            return readers[0] == reader || readers[1] == reader;
        }

        /// The chain that ends in this kind-1 node, as a descriptor for ek_hip_reduce_chain / ek_hip_map_chain: unevaluated maps
        /// below it that nobody else wants (unscaled, no live sincos partner) are absorbed, then an unevaluated arithmetic node
        /// (kind 4) that nobody else wants becomes the base; whatever cannot be absorbed is evaluated here and is the base.  The
        /// scale / partner of THIS node are its caller's business.
        void build_chain(ek_chain &ch, bool peek = false) {
            int ops[3], n = 0;
            ops[n++] = deferred->index_type;
            HIPBuffer *src = deferred->table, *below = this;        // below: the node of the chain that reads src
            while (n < 3 && src->absorbable_() && src->deferred->kind == 1 && !src->deferred->scaled &&
                   !(src->deferred->partner && src->deferred->partner->deferred)) {
                ops[n++] = src->deferred->index_type;
                below = src;
                src = src->deferred->table;
            }
            ch.n_maps = n;
            for (int k = 0; k < 3; ++k) ch.map_ops[k] = k < n ? ops[n - 1 - k] : (int) EK_COPY;        // first applied first
            if ((src->absorbable_() || (peek && src->only_sibling_maps_(below))) && src->deferred->kind == 4) {
                const Deferred *a = src->deferred;
                ch.arity = a->arity;
                ch.base_op = a->op;
                for (int k = 0; k < 3; ++k) {
                    HIPBuffer *b = k < a->arity ? a->operand_buf(k) : nullptr;
                    ch.src[k] = k >= a->arity ? ek_operand{ nullptr, 0, 0 }
                              : a->is_imm[k]  ? ek_operand{ nullptr, a->imm[k], 1 } : ek_operand{ b->ptr, 0, b->size };
                }
            } else {
                if (src->deferred) src->force();
                ch.arity = 1;
                ch.base_op = EK_COPY;
                ch.src[0] = ek_operand{ src->ptr, 0, src->size };
                ch.src[1] = ch.src[2] = ek_operand{ nullptr, 0, 0 };
            }
        }

        /// kind 3: zero<HIPArray>(n) that nobody has looked at yet.  The gradient buffers of the backward sweep start their life
        /// like this (autodiff.cpp:332-338 zero-fills them); a bucket-ordered scatter_add that takes one as its target WRITES
        /// its sums instead of adding them to a memset buffer (adopt_uninitialized()), everybody else gets the memset.
        void force_zeros() {
            void *p = nullptr;
            const size_t bytes = (size ? size : 1) * deferred->elem_size;
            hip_check(ek_hip_malloc(bytes, &p), "HIPArray (zeros)");
            if (ek_hip_memset(p, 0, bytes) != EK_OK) {
                ek_hip_free(p);
                hip_raise("HIPArray (zeros)");
            }
            ptr = p;
            drop_deferred();
        }
        /// Storage without contents for a pending zeros node whose only consumer is about to overwrite every entry
        void adopt_uninitialized() {
            void *p = nullptr;
            hip_check(ek_hip_malloc((size ? size : 1) * deferred->elem_size, &p), "HIPArray (zeros)");
            ptr = p;
            drop_deferred();
        }

        /// Execute the deferred gather / map
        void force() {
            if (!deferred) return;
            if (deferred->kind == 1) { force_map(); return; }
            if (deferred->kind == 2) { force_pair(); return; }
            if (deferred->kind == 3) { force_zeros(); return; }
            if (deferred->kind == 4) { force_arith(); return; }
            void *p = nullptr;
            hip_check(ek_hip_malloc((size ? size : 1) * deferred->elem_size, &p), "HIPArray (deferred gather)");
            ek_gathered g = gathered();
            int rc = ek_hip_gather(deferred->type, deferred->index_type, p, g.table, &g.index, &g.mask, size);
            if (rc != EK_OK) {
                ek_hip_free(p);
                hip_raise("HIPArray (deferred gather)");
            }
            ptr = p;
            drop_deferred();
        }

        /// Deferred gathers that read this buffer must run before its contents change
        void force_readers() {
            while (!readers.empty()) readers.back()->force();
            drop_narrowed();                   // (called whenever the contents may change)
        }

        void drop_deferred() {
            Deferred *d = deferred;
            deferred = nullptr;
            pending_unlink();
            if (d->partner && d->partner->deferred) d->partner->deferred->partner = nullptr;
            if (d->bucketed) ek_hip_bucketed_destroy(d->bucketed);
            for (HIPBuffer *src : { d->table, d->index, d->mask, d->table2, d->arg0 }) {
                if (!src) continue;
                auto &r = src->readers;
                for (size_t i = 0; i < r.size(); ++i)
                    if (r[i] == this) { r[i] = r.back(); r.pop_back(); break; }
            }
            unref(d->table);
            unref(d->index);
            unref(d->mask);
            unref(d->table2);
            unref(d->arg0);
            delete d;
        }

        void drop_host_mirror() {
            if (host_mirror) { free(host_mirror); host_mirror = nullptr; }
        }

        ~HIPBuffer() {
            if (deferred) drop_deferred();
            drop_narrowed();
            drop_host_mirror();
            if (owned && ptr) ek_hip_free(ptr);
            unref(view_of);
        }
    };

    /// Deferred evaluation (gathers and unary maps) can be switched off: ENOKI_HIP_DEFER=0 (or its first name,
    /// ENOKI_HIP_DEFER_GATHER=0), or hip_set_defer(false) / hip_set_defer_gather(false)
    /// Deferred evaluation (gathers and unary maps) can be switched off: ENOKI_HIP_DEFER=0 (or its first name,
    /// ENOKI_HIP_DEFER_GATHER=0), or hip_set_defer(false) / hip_set_defer_gather(false).  One switch per process, whichever
    /// shared object asks.
    inline bool &hip_defer_gather_flag() { return HIPBuffer::shared().defer; }
    inline bool hip_strict_safe_mul() {
        static const bool strict = [] { const char *e = getenv("ENOKI_HIP_STRICT_SAFE_MUL"); return e && *e == '1'; }();
        return strict;
    }

    /// Smallest array whose fusable unary results / gathers are left unevaluated (defaults 64 Ki / 4096 elements: below
    /// that a kernel launch costs more than the bytes it moves).  ENOKI_HIP_DEFER_MIN=<n> overrides both -- the test
    /// suites run with 1 so that every small tape program goes through the deferred paths.
    inline size_t hip_defer_min_override() {
        static const size_t value = [] {
            const char *e = getenv("ENOKI_HIP_DEFER_MIN");
            return e ? (size_t) strtoull(e, nullptr, 10) : (size_t) 0;
        }();
        return value;
    }
    template <typename T> struct hip_type;
    template <> struct hip_type<bool>     { static constexpr int value = EK_BOOL; };
    template <> struct hip_type<int32_t>  { static constexpr int value = EK_I32; };
    template <> struct hip_type<uint32_t> { static constexpr int value = EK_U32; };
    template <> struct hip_type<int64_t>  { static constexpr int value = EK_I64; };
    template <> struct hip_type<uint64_t> { static constexpr int value = EK_U64; };
    template <> struct hip_type<float>    { static constexpr int value = EK_F32; };
    template <> struct hip_type<double>   { static constexpr int value = EK_F64; };
}

namespace detail {
    /// Ops on two host-known scalars are evaluated on the host (single IEEE operations: same bits as the
    /// kernels); returns false for ops that need the device algorithms (transcendentals, ...).
    template <typename V> inline bool host_binary(int op, V a, V b, V &out) {
        if constexpr (std::is_floating_point_v<V>) {
            switch (op) {
                case EK_ADD: out = a + b; return true;
                case EK_SUB: out = a - b; return true;
                case EK_MUL: out = a * b; return true;
                case EK_DIV: out = a / b; return true;
                case EK_SAFE_MUL: out = (a == V(0) || b == V(0)) ? V(0) : a * b; return true;
                default: return false;
            }
        } else if constexpr (std::is_integral_v<V> && !std::is_same_v<V, bool>) {
            using U = std::make_unsigned_t<V>;
            switch (op) {
                case EK_ADD: out = (V) ((U) a + (U) b); return true;
                case EK_SUB: out = (V) ((U) a - (U) b); return true;
                case EK_MUL: out = (V) ((U) a * (U) b); return true;
                default: return false;
            }
        } else {
            return false;
        }
    }
}

template <typename Value_> struct HIPArray : ArrayTag {
    static_assert(std::is_arithmetic_v<Value_>, "HIPArray: arithmetic element types only");
    template <typename T> friend struct HIPArray;

    using Value = Value_;
    using Scalar = Value_;
    using ArrayType = HIPArray;
    using MaskType = HIPArray<bool>;
    template <typename T> using ReplaceValue = HIPArray<T>;
    template <typename T> using ReplaceScalar = HIPArray<T>;
    template <typename T> using ReplaceMaskValue = HIPArray<T>;

    static constexpr int Type = detail::hip_type<Value>::value;
    static constexpr size_t Depth = 1;
    static constexpr size_t Rank = 2;
    static constexpr bool IsMask = std::is_same_v<Value, bool>;
    static constexpr bool IsDiff = false;
    static constexpr bool IsDynamic = true;
    static constexpr bool IsDevice = true;
    static constexpr bool IsCUDA = false;
    static constexpr bool IsFloat = std::is_floating_point_v<Value>;
    static constexpr bool IsInt = std::is_integral_v<Value> && !IsMask;

    // -----------------------------------------------------------------------------------------
    //  Construction, assignment
    // -----------------------------------------------------------------------------------------

    HIPArray() = default;

    ~HIPArray() { release(); }

    HIPArray(const HIPArray &a) : m_buf(a.m_buf), m_imm(a.m_imm), m_is_imm(a.m_is_imm) {
        if (m_buf) m_buf->ref_count++;
    }

    HIPArray(HIPArray &&a) noexcept : m_buf(a.m_buf), m_imm(a.m_imm), m_is_imm(a.m_is_imm) {
        a.m_buf = nullptr;
        a.m_is_imm = false;
    }

    /// Broadcast a host scalar (an immediate: no device memory is touched)
    HIPArray(Value value) : m_imm(value), m_is_imm(true) { }

    template <typename T, enable_if_t<std::is_arithmetic_v<T> && !std::is_same_v<T, Value>> = 0>
    HIPArray(T value) : m_imm((Value) value), m_is_imm(true) { }

    /// Element list -> device (cuda.h:319-323)
    template <typename... Args, enable_if_t<(sizeof...(Args) > 1) && (std::is_arithmetic_v<Args> && ...)> = 0>
    HIPArray(Args... args) {
        Value data[] = { (Value) args... };
        *this = copy(data, sizeof...(Args));
    }

    /// Converting constructor (cuda.h:236-247): float->int truncates, int->float rounds to nearest
    template <typename T, enable_if_t<!std::is_same_v<T, Value>> = 0>
    HIPArray(const HIPArray<T> &v) {
        if (v.m_is_imm) {
            m_imm = (Value) v.m_imm;
            m_is_imm = true;
        } else if (v.m_buf) {
            size_t n = v.size();
            // 64-bit integers narrowed to 32 bits: kept on the source (detail::HIPBuffer::narrowed)
            constexpr bool Narrowing = std::is_integral_v<T> && sizeof(T) == 8 && std::is_integral_v<Value> && sizeof(Value) == 4 && !IsMask;
            if constexpr (Narrowing) {
                if (v.m_buf->narrowed && v.m_buf->narrowed_type == Type && !v.m_buf->narrowed->exported && !v.m_buf->exported) {
                    m_buf = v.m_buf->narrowed;
                    m_buf->ref_count++;
                    return;
                }
            }
            allocate(n);
            ek_operand oa = v.operand();
            detail::hip_check(ek_hip_cast(HIPArray<T>::Type, Type, m_buf->ptr, &oa, n), "HIPArray(cast)");
            if constexpr (Narrowing) {
                if (n >= 4096 && v.m_buf->owned && !v.m_buf->exported && !v.m_buf->deferred) {
                    v.m_buf->drop_narrowed();
                    v.m_buf->narrowed = m_buf;
                    v.m_buf->narrowed_type = Type;
                    m_buf->narrowed_of = v.m_buf;
                    m_buf->ref_count++;
                }
            }
        }
    }

    /// Reinterpreting constructor (cuda.h:249-258): shares the buffer
    template <typename T, enable_if_t<!std::is_same_v<T, Value>> = 0>
    HIPArray(const HIPArray<T> &v, detail::reinterpret_flag) {
        static_assert(sizeof(T) == sizeof(Value), "reinterpret_array(): element sizes must match");
        if (v.m_is_imm) {
            memcpy(&m_imm, &v.m_imm, sizeof(Value));
            m_is_imm = true;
        } else if (v.m_buf) {
            m_buf = v.m_buf;
            m_buf->ref_count++;
        }
    }

    HIPArray(const HIPArray &v, detail::reinterpret_flag) : HIPArray(v) { }

    HIPArray &operator=(const HIPArray &a) {
        detail::HIPBuffer *buf = a.m_buf;       // `a` may be *this: read it before this handle lets go
        if (buf) buf->ref_count++;
        release();
        m_buf = buf;
        m_imm = a.m_imm;
        m_is_imm = a.m_is_imm;
        return *this;
    }

    HIPArray &operator=(HIPArray &&a) noexcept {
        std::swap(m_buf, a.m_buf);
        std::swap(m_imm, a.m_imm);
        std::swap(m_is_imm, a.m_is_imm);
        return *this;
    }

    // -----------------------------------------------------------------------------------------
    //  Vertical operations
    // -----------------------------------------------------------------------------------------

    HIPArray add_(const HIPArray &v) const { return binary(EK_ADD, v, "add_"); }
    HIPArray sub_(const HIPArray &v) const { return binary(EK_SUB, v, "sub_"); }
    HIPArray mul_(const HIPArray &v) const { return binary(EK_MUL, v, "mul_"); }
    HIPArray div_(const HIPArray &v) const { return binary(EK_DIV, v, "div_"); }
    HIPArray mod_(const HIPArray &v) const { return binary(EK_MOD, v, "mod_"); }
    HIPArray mulhi_(const HIPArray &v) const { return binary(EK_MULHI, v, "mulhi_"); }
    HIPArray min_(const HIPArray &v) const { return binary(EK_MIN, v, "min_"); }
    HIPArray max_(const HIPArray &v) const { return binary(EK_MAX, v, "max_"); }
    HIPArray xor_(const HIPArray &v) const { return binary(EK_XOR, v, "xor_"); }
    HIPArray sl_(const HIPArray &v) const { return binary(EK_SL, v, "sl_"); }
    HIPArray sr_(const HIPArray &v) const { return binary(EK_SR, v, "sr_"); }
    HIPArray sl_(size_t k) const { return sl_(HIPArray((Value) k)); }
    HIPArray sr_(size_t k) const { return sr_(HIPArray((Value) k)); }
    template <size_t Imm> HIPArray sl_() const { return sl_(Imm); }
    template <size_t Imm> HIPArray sr_() const { return sr_(Imm); }

    HIPArray fmadd_(const HIPArray &b, const HIPArray &c) const { return ternary(EK_FMADD, b, c, "fmadd_"); }
    HIPArray fmsub_(const HIPArray &b, const HIPArray &c) const { return ternary(EK_FMSUB, b, c, "fmsub_"); }
    HIPArray fnmadd_(const HIPArray &b, const HIPArray &c) const { return ternary(EK_FNMADD, b, c, "fnmadd_"); }
    HIPArray fnmsub_(const HIPArray &b, const HIPArray &c) const { return ternary(EK_FNMSUB, b, c, "fnmsub_"); }

    HIPArray neg_() const { return unary(EK_NEG, "neg_"); }
    HIPArray abs_() const { return unary(EK_ABS, "abs_"); }
    HIPArray not_() const { return unary(EK_NOT, "not_"); }
    HIPArray sqrt_() const { return unary(EK_SQRT, "sqrt_"); }
    HIPArray rcp_() const { return unary(EK_RCP, "rcp_"); }
    HIPArray rsqrt_() const { return unary(EK_RSQRT, "rsqrt_"); }
    HIPArray floor_() const { return unary(EK_FLOOR, "floor_"); }
    HIPArray ceil_() const { return unary(EK_CEIL, "ceil_"); }
    HIPArray round_() const { return unary(EK_ROUND, "round_"); }
    HIPArray trunc_() const { return unary(EK_TRUNC, "trunc_"); }
    HIPArray sin_() const { return unary(EK_SIN, "sin_"); }
    HIPArray cos_() const { return unary(EK_COS, "cos_"); }
    HIPArray exp_() const { return unary(EK_EXP, "exp_"); }
    HIPArray log_() const { return unary(EK_LOG, "log_"); }
    HIPArray popcnt_() const { return unary(EK_POPCNT, "popcnt_"); }
    HIPArray lzcnt_() const { return unary(EK_LZCNT, "lzcnt_"); }
    HIPArray tzcnt_() const { return unary(EK_TZCNT, "tzcnt_"); }
    HIPArray sign_() const { return unary(EK_SIGN, "sign_"); }

    template <typename T> T floor2int_() const { return T(floor_()); }
    template <typename T> T ceil2int_() const { return T(ceil_()); }

    // second wave (array_math.h:466-1348): one fused kernel each, f32
    HIPArray tan_() const { return unary(EK_TAN, "tan_"); }
    HIPArray cot_() const { return unary(EK_COT, "cot_"); }
    HIPArray asin_() const { return unary(EK_ASIN, "asin_"); }
    HIPArray acos_() const { return unary(EK_ACOS, "acos_"); }
    HIPArray atan_() const { return unary(EK_ATAN, "atan_"); }
    HIPArray sinh_() const { return unary(EK_SINH, "sinh_"); }
    HIPArray cosh_() const { return unary(EK_COSH, "cosh_"); }
    HIPArray tanh_() const { return unary(EK_TANH, "tanh_"); }
    /// the derivative weights of tan, tanh, atan as one op of the argument (same roundings as sqr(sec(x)), sqr(sech(x)), rcp(1 + sqr(x)))
    HIPArray sec_sqr_() const { return unary(EK_SEC_SQR, "sec_sqr_"); }
    HIPArray sech_sqr_() const { return unary(EK_SECH_SQR, "sech_sqr_"); }
    HIPArray rcp_1p_sqr_() const { return unary(EK_RCP_1P_SQR, "rcp_1p_sqr_"); }
    HIPArray asinh_() const { return unary(EK_ASINH, "asinh_"); }
    HIPArray acosh_() const { return unary(EK_ACOSH, "acosh_"); }
    HIPArray atanh_() const { return unary(EK_ATANH, "atanh_"); }
    HIPArray cbrt_() const { return unary(EK_CBRT, "cbrt_"); }
    // special functions (enoki/special.h; reference include/enoki/special.h:56-312), one fused kernel each
    HIPArray erf_() const { return unary(EK_ERF, "erf_"); }
    HIPArray erfc_() const { return unary(EK_ERFC, "erfc_"); }
    HIPArray erfinv_() const { return unary(EK_ERFINV, "erfinv_"); }
    HIPArray i0e_() const { return unary(EK_I0E, "i0e_"); }
    HIPArray dawson_() const { return unary(EK_DAWSON, "dawson_"); }
    HIPArray erfi_() const { return unary(EK_ERFI, "erfi_"); }
    HIPArray lgamma_() const { return unary(EK_LGAMMA, "lgamma_"); }
    HIPArray tgamma_() const { return unary(EK_TGAMMA, "tgamma_"); }
    HIPArray atan2_(const HIPArray &x) const { return binary(EK_ATAN2, x, "atan2_"); }
    HIPArray pow_(const HIPArray &y) const { return binary(EK_POW, y, "pow_"); }
    HIPArray fmod_(const HIPArray &y) const { return binary(EK_FMOD, y, "fmod_"); }
    HIPArray ldexp_(const HIPArray &e) const { return binary(EK_LDEXP, e, "ldexp_"); }

    std::pair<HIPArray, HIPArray> sincosh_() const {
        require_valid("sincosh_");
        size_t n = size();
        HIPArray s = empty_(n), c = empty_(n);
        ek_operand oa = operand();
        detail::hip_check(ek_hip_sincosh(Type, s.m_buf->ptr, c.m_buf->ptr, &oa, n), "sincosh_");
        return { std::move(s), std::move(c) };
    }

    /// Both results from one pass over the input
    std::pair<HIPArray, HIPArray> sincos_() const {
        require_valid("sincos_");
        if constexpr (IsFloat) {
            if (can_defer_map_()) {
                // both halves unevaluated and linked: whichever is materialised first fills both with ONE sincos kernel;
                // a half that is only ever consumed on load (hsum(sin(x)), the cos(x) of the adjoint) is never written
                HIPArray s = defer_map_(EK_SIN), c = defer_map_(EK_COS);
                s.m_buf->deferred->partner = c.m_buf;
                c.m_buf->deferred->partner = s.m_buf;
                return { std::move(s), std::move(c) };
            }
        }
        size_t n = size();
        HIPArray s = empty_(n), c = empty_(n);
        ek_operand oa = operand();
        detail::hip_check(ek_hip_sincos(Type, s.m_buf->ptr, c.m_buf->ptr, &oa, n), "sincos_");
        return { std::move(s), std::move(c) };
    }

    /// and_/or_ with an operand of the same type are bit operations; with a mask they select
    /// (cuda.h:545-575)
    HIPArray and_(const HIPArray &v) const { return binary(EK_AND, v, "and_"); }
    HIPArray or_(const HIPArray &v) const { return binary(EK_OR, v, "or_"); }

    template <typename T = Value, enable_if_t<!std::is_same_v<T, bool>> = 0>
    HIPArray and_(const MaskType &m) const { return select_(m, *this, HIPArray(Value(0))); }

    template <typename T = Value, enable_if_t<!std::is_same_v<T, bool>> = 0>
    HIPArray or_(const MaskType &m) const {
        using UInt = scalar_t<uint_array_t<HIPArray>>;
        UInt ones = (UInt) -1;
        Value all_ones;
        memcpy(&all_ones, &ones, sizeof(Value));
        return select_(m, HIPArray(all_ones), *this);
    }

    template <typename T> HIPArray andnot_(const HIPArray<T> &v) const { return and_(v.not_()); }

    MaskType eq_(const HIPArray &v) const { return compare(EK_EQ, v, "eq_"); }
    MaskType neq_(const HIPArray &v) const { return compare(EK_NEQ, v, "neq_"); }
    MaskType lt_(const HIPArray &v) const { return compare(EK_LT, v, "lt_"); }
    MaskType le_(const HIPArray &v) const { return compare(EK_LE, v, "le_"); }
    MaskType gt_(const HIPArray &v) const { return compare(EK_GT, v, "gt_"); }
    MaskType ge_(const HIPArray &v) const { return compare(EK_GE, v, "ge_"); }

    static HIPArray select_(const MaskType &m, const HIPArray &t, const HIPArray &f) {
        m.require_valid("select_"); t.require_valid("select_"); f.require_valid("select_");
        size_t n = broadcast_size(broadcast_size(m.size(), t.size()), f.size());
        HIPArray r = empty_(n);
        ek_operand om = m.operand(), ot = t.operand(), of = f.operand();
        detail::hip_check(ek_hip_select(Type, r.m_buf->ptr, &om, &ot, &of, n), "select_");
        return r;
    }

    // -----------------------------------------------------------------------------------------
    //  Fused primitives of the reverse/forward sweeps (src/autodiff/autodiff.cpp:1191-1221)
    // -----------------------------------------------------------------------------------------

    /// (w == 0 || g == 0) ? 0 : w * g
    static HIPArray safe_mul_(const HIPArray &w, const HIPArray &g) { return w.binary(EK_SAFE_MUL, g, "safe_mul"); }

    /// (w == 0 || g == 0) ? acc : fma(w, g, acc)
    static HIPArray safe_fmadd_(const HIPArray &w, const HIPArray &g, const HIPArray &acc) {
        return w.ternary(EK_SAFE_FMADD, g, acc, "safe_fmadd");
    }

    /// hsum(safe_mul(w, g)) in one pass
    static HIPArray hsum_safe_mul_(const HIPArray &w, const HIPArray &g) {
        w.require_valid("hsum_safe_mul"); g.require_valid("hsum_safe_mul");
        size_t n = broadcast_size(w.size(), g.size());
        HIPArray r = empty_(1);
        ek_operand ow = w.operand(), og = g.operand();
        detail::hip_check(ek_hip_hsum_safe_mul(Type, r.m_buf->ptr, &ow, &og, n), "hsum_safe_mul");
        return r;
    }

    // -----------------------------------------------------------------------------------------
    //  Initialization
    // -----------------------------------------------------------------------------------------

    static HIPArray empty_(size_t size) {
        HIPArray r;
        r.allocate(size);
        return r;
    }

    static HIPArray zero_(size_t size) {
        if (size == 1) return HIPArray(Value(0));
        if constexpr (IsFloat) {
            // left unevaluated (kind 3, see detail::HIPBuffer::force_zeros): the usual fate of a zero array in the backward
            // sweep is to become the target of ONE scatter_add, which can then write instead of accumulate
            const size_t least = detail::hip_defer_min_override() ? detail::hip_defer_min_override() : defer_min_size_;
            if (detail::hip_defer_gather_flag() && size >= least) {
                auto *d = new typename detail::HIPBuffer::Deferred{ nullptr, nullptr, nullptr, Type, 0, sizeof(Value), false, 3, nullptr };
                HIPArray r;
                r.m_buf = new detail::HIPBuffer();
                r.m_buf->size = size;
                r.m_buf->deferred = d;
                r.m_buf->pending_link();
                return r;
            }
        }
        HIPArray r = empty_(size);
        if (size) detail::hip_check(ek_hip_memset(r.m_buf->ptr, 0, size * sizeof(Value)), "zero_");
        return r;
    }

    static HIPArray full_(const Value &value, size_t size) {
        if (size == 1) return HIPArray(value);
        HIPArray r = empty_(size);
        detail::hip_check(ek_hip_fill(Type, r.m_buf->ptr, imm_bits(value), size), "full_");
        return r;
    }

    static HIPArray arange_(ptrdiff_t start, ptrdiff_t stop, ptrdiff_t step) {
        size_t size = size_t((stop - start + step - (step > 0 ? 1 : -1)) / step);
        HIPArray r = empty_(size);
        detail::hip_check(ek_hip_arange(Type, r.m_buf->ptr, (int64_t) start, (int64_t) step, size), "arange_");
        return r;
    }

    static HIPArray linspace_(Value min, Value max, size_t size) {
        HIPArray r = empty_(size);
        detail::hip_check(ek_hip_linspace(Type, r.m_buf->ptr, (double) min, (double) max, size), "linspace_");
        return r;
    }

    /// Wrap existing device memory (cuda.h:796-798); `dealloc`: pointer came from ek_hip_malloc
    static HIPArray map(void *ptr, size_t size, bool dealloc = false) {
        HIPArray r;
        r.m_buf = new detail::HIPBuffer();
        r.m_buf->ptr = ptr;
        r.m_buf->size = size;
        r.m_buf->owned = dealloc;
        return r;
    }

    /// A read-only window [begin, begin + size) into `parent` that shares ownership of the parent's storage: it stays valid for
    /// as long as any window (or the parent) is alive.  (The groups that partition() hands out are such windows into ONE sorted
    /// lane array: a group copied out of the partition may outlive the pointer array it came from.)
    static HIPArray view_(const HIPArray &parent, size_t begin, size_t size) {
        HIPArray r = map((void *) (parent.data() + begin), size, false);
        r.m_buf->view_of = parent.m_buf;
        parent.m_buf->ref_count++;
        return r;
    }

    /// Copy host memory to a new device array (cuda.h:800-802)
    static HIPArray copy(const void *ptr, size_t size) {
        HIPArray r = empty_(size);
        if (size)
            detail::hip_check(ek_hip_memcpy_to_device(r.m_buf->ptr, ptr, size * sizeof(Value)), "copy");
        return r;
    }

    // -----------------------------------------------------------------------------------------
    //  Indexed memory operations (cuda.h:845-905)
    // -----------------------------------------------------------------------------------------

    template <size_t Stride, typename Index>
    static HIPArray gather_(const void *ptr, const Index &index, const MaskType &mask) {
        static_assert(Stride == sizeof(Value) || Stride == 0, "HIPArray::gather_(): element stride only");
        index.require_valid("gather_"); mask.require_valid("gather_");
        size_t n = broadcast_size(index.size(), mask.size());
        HIPArray r = empty_(n);
        ek_operand oi = index.operand(), om = mask.operand();
        detail::hip_check(ek_hip_gather(Type, Index::Type, r.m_buf->ptr, ptr, &oi, &om, n), "gather_");
        return r;
    }

    template <size_t Stride, typename Index>
    void scatter_(void *ptr, const Index &index, const MaskType &mask) const {
        static_assert(Stride == sizeof(Value) || Stride == 0, "HIPArray::scatter_(): element stride only");
        require_valid("scatter_"); index.require_valid("scatter_"); mask.require_valid("scatter_");
        size_t n = broadcast_size(broadcast_size(size(), index.size()), mask.size());
        ek_operand ov = operand(), oi = index.operand(), om = mask.operand();
        detail::hip_check(ek_hip_scatter(Type, Index::Type, ptr, &ov, &oi, &om, n), "scatter_");
    }

    template <size_t Stride, typename Index>
    void scatter_add_(void *ptr, const Index &index, const MaskType &mask, size_t target_size = 0) const {
        static_assert(Stride == sizeof(Value) || Stride == 0, "HIPArray::scatter_add_(): element stride only");
        require_valid("scatter_add_"); index.require_valid("scatter_add_"); mask.require_valid("scatter_add_");
        size_t n = broadcast_size(broadcast_size(size(), index.size()), mask.size());
        ek_operand ov = operand(), oi = index.operand(), om = mask.operand();
        detail::hip_check(ek_hip_scatter_add(Type, Index::Type, ptr, target_size, &ov, &oi, &om, n, 0), "scatter_add_");
    }

    /// out[i] = mask[i] && address[i] ? *(Value *) (address[i] + byte_offset) : 0 (ENOKI_CALL_SUPPORT_GETTER on the device)
    static HIPArray gather_address_(const HIPArray<uint64_t> &address, ptrdiff_t byte_offset, const MaskType &mask) {
        address.require_valid("gather_address_"); mask.require_valid("gather_address_");
        size_t n = broadcast_size(address.size(), mask.size());
        HIPArray r = empty_(n);
        ek_operand oa = address.operand(), om = mask.operand();
        detail::hip_check(ek_hip_gather_address(Type, r.m_buf->ptr, &oa, (int64_t) byte_offset, &om, n), "gather_address_");
        return r;
    }

    /// gather/scatter with an array as source/target (array_struct.h:9-123); a source of size 1 is
    /// a broadcast (array_struct.h:19-22)
    template <bool IsPermute, typename Index>
    static HIPArray gather_array_(const HIPArray &source, const Index &index, const MaskType &mask) {
        if (source.size() <= 1) return source & mask;
        if constexpr (IsFloat && (Index::Type == EK_U32 || Index::Type == EK_I32)) {
            if (HIPArray r = defer_gather_(source, detach(index), mask); r.valid()) return r;
        }
        if constexpr (IsFloat && (Index::Type == EK_U64 || Index::Type == EK_I64)) {
            // a table has fewer than 2^32 entries: valid 64-bit indices are narrowed ONCE (the result stays on the index array)
            // and take the 32-bit paths -- deferral, bucket order -- like the tape's own offset arrays
            const auto &ix = detach(index);
            if (source.size() <= ((size_t) 1 << 32) && !ix.m_is_imm && ix.m_buf && ix.m_buf->size >= 4096) {
                HIPArray<uint32_t> narrow(ix);
                if (HIPArray r = defer_gather_(source, narrow, mask); r.valid()) return r;
                return gather_<sizeof(Value)>(source.data(), narrow, mask);
            }
        }
        return gather_<sizeof(Value)>(source.data(), detach(index), mask);
    }

    /// Large gathers from small tables are deferred (see detail::HIPBuffer); returns an invalid array when this one is not
    static constexpr size_t defer_min_size_ = 4096, defer_max_table_bytes_ = (size_t) 512 << 20;
    template <typename Index>
    static HIPArray defer_gather_(const HIPArray &source, const Index &index, const MaskType &mask) {
        HIPArray r;
        // tables that an external view may write behind our back (zero-copy exports), index / mask arrays in foreign memory:
        // evaluated right away
        if (!detail::hip_defer_gather_flag() || !source.m_buf || !source.m_buf->owned || source.m_buf->exported ||
            !index.m_buf || !index.m_buf->owned || index.m_buf->exported)
            return r;
        const size_t n = index.m_buf->size;
        const size_t least = detail::hip_defer_min_override() ? detail::hip_defer_min_override() : defer_min_size_;
        if (n < least || n < 2 || source.m_buf->size * sizeof(Value) > defer_max_table_bytes_) return r;
        if (mask.m_is_imm ? !mask.m_imm : (!mask.m_buf || mask.m_buf->size != n || !mask.m_buf->owned || mask.m_buf->exported)) return r;
        source.ptr_();                                       // a table that is itself deferred runs first
        auto *d = new typename detail::HIPBuffer::Deferred{ source.m_buf, index.m_buf, mask.m_is_imm ? nullptr : mask.m_buf,
                                                            Type, Index::Type, sizeof(Value), false };
        d->table->ref_count++;
        d->index->ref_count++;
        if (d->mask) d->mask->ref_count++;
        r.m_buf = new detail::HIPBuffer();
        r.m_buf->size = n;
        r.m_buf->deferred = d;
        r.m_buf->pending_link();
        // every buffer the node will read hands out mutable pointers only after the node has run (data(), make_unique())
        d->table->readers.push_back(r.m_buf);
        if (d->index != d->table) d->index->readers.push_back(r.m_buf);
        if (d->mask && d->mask != d->table && d->mask != d->index) d->mask->readers.push_back(r.m_buf);
        return r;
    }

    bool deferred_() const { return m_buf && m_buf->deferred && m_buf->deferred->kind == 0 && !m_buf->deferred->consumed; }
    /// An unevaluated unary map (see detail::HIPBuffer)
    bool mapped_() const { return m_buf && m_buf->deferred && m_buf->deferred->kind == 1; }
    /// zero<HIPArray>(n) that has not been looked at yet (kind 3)
    bool zeroed_() const { return m_buf && m_buf->deferred && m_buf->deferred->kind == 3; }
    /// An unevaluated fma over a gathered pair (kind 2, see detail::HIPBuffer)
    bool paired_() const { return m_buf && m_buf->deferred && m_buf->deferred->kind == 2; }

    /// gather(A, idx) * x that has not run yet (kind 2 without an addend table, see detail::HIPBuffer)
    bool gathered_product_() const { return m_buf && m_buf->deferred && m_buf->deferred->kind == 2 && !m_buf->deferred->table2; }

    /// g * x with g an unevaluated gather and x an evaluated array of the same length: left unevaluated (kind 2 without an addend)
    /// when the library's bucket-ordered path covers the shape.  Invalid array: not that shape -- the caller consumes the gather
    /// in place right away.
    static HIPArray defer_gathered_product_(const HIPArray &g, const HIPArray &x, size_t n) {
        HIPArray r;
        if constexpr (IsFloat) {
            const auto *p = g.m_buf->deferred;
            if (!detail::hip_defer_gather_flag() || x.m_is_imm || !x.m_buf || x.m_buf->size != n || !x.m_buf->owned ||
                x.m_buf->exported || x.m_buf->deferred || g.m_buf->size != n)
                return r;
            if (p->mask && sizeof(Value) != 4) return r;
            if (!ek_hip_bucketed_applicable(Type, p->index_type, p->table->size, n)) return r;
            auto *d = new typename detail::HIPBuffer::Deferred{ p->table, p->index, p->mask, Type, p->index_type, sizeof(Value),
                                                                false, 2, nullptr };
            d->arg0 = x.m_buf;
            d->op = EK_MULADD;
            r.m_buf = new detail::HIPBuffer();
            r.m_buf->size = n;
            r.m_buf->deferred = d;
            r.m_buf->pending_link();
            detail::HIPBuffer *seen[4] = { nullptr, nullptr, nullptr, nullptr };
            int k = 0;
            for (detail::HIPBuffer *src : { d->table, d->index, d->arg0, d->mask }) {
                if (!src) continue;
                src->ref_count++;
                bool dup = false;
                for (int j = 0; j < k; ++j) dup = dup || seen[j] == src;
                if (!dup) { src->readers.push_back(r.m_buf); seen[k++] = src; }
            }
            g.m_buf->deferred->consumed = true;
        }
        return r;
    }

    /// (this = gather(A, idx) * x, unevaluated)  +/-  gc = gather(C, idx)  through the same index / mask array and equally sized
    /// tables:  the kind-2 node of the whole expression with a product-then-sum op.  form: EK_MULADD (p + gc), EK_MULSUB (p - gc),
    /// EK_NMULADD (gc - p).  Invalid array: not that shape.
    HIPArray gathered_product_plus_(int form, const HIPArray &gc, size_t n) const {
        HIPArray r;
        if constexpr (IsFloat) {
            const auto *p = m_buf->deferred, *q = gc.m_buf->deferred;
            if (p->index != q->index || p->mask != q->mask || p->index_type != q->index_type || p->table->size != q->table->size ||
                m_buf->size != n || gc.m_buf->size != n)
                return r;
            HIPArray x;
            x.m_buf = p->arg0;
            x.m_buf->ref_count++;
            r = defer_pair_fma_(form, *this, x, gc, n);
        }
        return r;
    }

    /// An unevaluated arithmetic op over evaluated operands (kind 4, see detail::HIPBuffer)
    bool arith_() const { return m_buf && m_buf->deferred && m_buf->deferred->kind == 4; }

    /// op(x[0], .., x[arity - 1]) left unevaluated (kind 4): every operand a host scalar or an evaluated, owned array of n elements
    /// that no external view can write.  Invalid array: not such a shape -- the caller runs the kernel.
    static HIPArray defer_arith_(int arity, int op, const HIPArray *const *x, size_t n) {
        HIPArray r;
        if constexpr (IsFloat) {
            const size_t least = detail::hip_defer_min_override() ? detail::hip_defer_min_override() : defer_map_min_size_;
            if (!detail::hip_defer_gather_flag() || n < least || n < 2) return r;
            bool any_array = false;
            for (int k = 0; k < arity; ++k) {
                if (x[k]->m_is_imm) continue;
                const detail::HIPBuffer *b = x[k]->m_buf;
                if (!b || b->size != n || !b->owned || b->exported || b->deferred) return r;
                any_array = true;
            }
            if (!any_array) return r;
            auto *d = new typename detail::HIPBuffer::Deferred{ nullptr, nullptr, nullptr, Type, 0, sizeof(Value), false, 4, nullptr };
            d->arity = arity;
            d->op = op;
            r.m_buf = new detail::HIPBuffer();
            r.m_buf->size = n;
            r.m_buf->deferred = d;
            r.m_buf->pending_link();
            for (int k = 0; k < arity; ++k) {
                if (x[k]->m_is_imm) { d->is_imm[k] = true; d->imm[k] = imm_bits(x[k]->m_imm); continue; }
                detail::HIPBuffer *b = x[k]->m_buf;
                (k == 0 ? d->table : k == 1 ? d->arg0 : d->table2) = b;
                b->ref_count++;
                bool dup = false;
                for (int j = 0; j < k; ++j) dup = dup || (!x[j]->m_is_imm && x[j]->m_buf == b);
                if (!dup) b->readers.push_back(r.m_buf);
            }
        }
        return r;
    }

    /// this (an unevaluated PRODUCT p * q, kind 4) plus / minus `other`, or `other` minus it, as ONE unevaluated node
    /// (EK_MULADD / EK_MULSUB / EK_NMULADD: the same two roundings) -- `a * x + b` written with operators.  The product node
    /// itself stays as it is (whoever else holds it can still evaluate it).  Invalid array: not that shape.
    HIPArray product_plus_(int form, const HIPArray &other, size_t n) const {
        HIPArray r;
        if constexpr (IsFloat) {
            const auto *d = m_buf->deferred;
            if (d->op != EK_MUL || d->arity != 2 || m_buf->size != n) return r;
            HIPArray p, q;
            auto operand_of = [&](int k) {
                HIPArray a;
                if (d->is_imm[k]) { Value v; memcpy(&v, &d->imm[k], sizeof(Value)); a = HIPArray(v); }
                else { a.m_buf = d->operand_buf(k); a.m_buf->ref_count++; }
                return a;
            };
            p = operand_of(0); q = operand_of(1);
            const HIPArray *x[3] = { &p, &q, &other };
            r = defer_arith_(3, form, x, n);
        }
        return r;
    }

    /// Unary ops whose result is left unevaluated until its first consumer: the ones ek_hip_reduce_map /
    /// ek_hip_scatter_add_multi_map can apply on load.  Small arrays are evaluated right away (nothing to win).
    static constexpr size_t defer_map_min_size_ = (size_t) 1 << 16;
    /// what EVERY streaming consumer applies on load: reductions, chains, the value streams of scatter_add, the bucket-ordered kernels
    static constexpr bool map_on_load_(int op) {
        return op == EK_NEG || op == EK_ABS || op == EK_SQRT || op == EK_RCP || op == EK_RSQRT || op == EK_SIN ||
               op == EK_COS || op == EK_EXP || op == EK_LOG || op == EK_RCP_SQR || op == EK_RSQRT_SQR || op == EK_RSQRT_CUBE;
    }
    /// ... and what is left unevaluated: those, plus (round 6) the second-wave functions whose derivative is ONE map of the same argument
    /// -- tan, tanh, atan, sinh, cosh -- and those derivative maps (EK_SEC_SQR, EK_SECH_SQR, EK_RCP_1P_SQR): reductions and chains apply
    /// them on load (ek_hip_reduce_map / ek_hip_reduce_chain / ek_hip_map_chain), every other consumer evaluates the map first
    static constexpr bool map_fusable_(int op) {
        return map_on_load_(op) || op == EK_TAN || op == EK_TANH || op == EK_ATAN || op == EK_SINH || op == EK_COSH ||
               op == EK_SEC_SQR || op == EK_SECH_SQR || op == EK_RCP_1P_SQR;
    }
    bool can_defer_map_() const {
        const size_t least = detail::hip_defer_min_override() ? detail::hip_defer_min_override() : defer_map_min_size_;
        return IsFloat && detail::hip_defer_gather_flag() && !m_is_imm && m_buf && m_buf->owned && !m_buf->exported &&
               m_buf->size >= least && m_buf->size > 1;
    }
    HIPArray defer_map_(int op) const {
        // A source that is a deferred gather runs first.  An fma of gathers stays (its consumer may want it in bucket order), and
        // so do an unevaluated arithmetic node and up to two levels of unevaluated maps: a reduction that consumes the chain
        // applies all of it while it loads the operands (detail::HIPBuffer::build_chain).
        bool keep = paired_() || arith_();
        if (mapped_()) {
            int depth = 1;
            for (const detail::HIPBuffer *b = m_buf->deferred->table; b->deferred && b->deferred->kind == 1; b = b->deferred->table) ++depth;
            keep = depth <= 2;
        }
        if (!keep) ptr_();
        auto *d = new typename detail::HIPBuffer::Deferred{ m_buf, nullptr, nullptr, Type, op, sizeof(Value), false, 1, nullptr };
        m_buf->ref_count++;
        HIPArray r;
        r.m_buf = new detail::HIPBuffer();
        r.m_buf->size = m_buf->size;
        r.m_buf->deferred = d;
        r.m_buf->pending_link();
        m_buf->readers.push_back(r.m_buf);
        return r;
    }

    /// factor * (this unevaluated map) as another unevaluated map of the same source (detail::HIPBuffer::Deferred::scale_bits).
    /// Invalid array when the product would not have the bits of the eager evaluation: two inexact factors in a row
    /// ((f c1) c2 rounds twice, f (c1 c2) once), a factor that is not finite, deferral switched off.
    HIPArray scaled_map_(Value factor) const {
        HIPArray r;
        if constexpr (IsFloat) {
            const auto *d = m_buf->deferred;
            if (!detail::hip_defer_gather_flag() || !(factor == factor) || factor - factor != Value(0)) return r;
            if (factor == Value(1)) return *this;          // the same function of the same source: the same node (x * 1 is x)
            Value scale = factor;
            if (d->scaled) {
                Value old;
                memcpy(&old, &d->scale_bits, sizeof(Value));
                // (c1 f) c2 == f (c1 c2) bit for bit when one of the factors is +-1 -- or, for rsqrt (whose values stay
                // clear of the denormal range), when the first one is a power of two: the .5 rsqrt(u) that d/du sqrt(u) records
                // times the seed of backward(c y)
                int e2 = 0;
                const bool pow2 = d->index_type == EK_RSQRT && std::fabs(std::frexp(old, &e2)) == Value(0.5) && e2 >= -32 && e2 <= 32;
                if (old != Value(1) && old != Value(-1) && factor != Value(1) && factor != Value(-1) && !pow2) return r;
                scale = old * factor;                 // exact
            }
            detail::HIPBuffer *src = d->table;
            auto *nd = new typename detail::HIPBuffer::Deferred{ src, nullptr, nullptr, Type, d->index_type, sizeof(Value), false, 1, nullptr };
            nd->scaled = true;
            nd->scale_bits = imm_bits(scale);
            src->ref_count++;
            r.m_buf = new detail::HIPBuffer();
            r.m_buf->size = m_buf->size;
            r.m_buf->deferred = nd;
            r.m_buf->pending_link();
            src->readers.push_back(r.m_buf);
        }
        return r;
    }
    /// c / sqrt(u) with sqrt(u) (this) an unevaluated map and c a power of two -- the weight `.5f / result` that d/du sqrt(u)
    /// records (autodiff.h:353-364) -- as the unevaluated map c * rsqrt(u) of the same source: rsqrt is 1 / sqrt(u) with both
    /// operations correctly rounded, and scaling by a power of two commutes with the rounding of the division (no result of
    /// 1 / sqrt(u) is near the denormal range), so the bits are those of the eager division.  The derivative of sqrt then is
    /// a function of u the bucket-ordered consumers can form themselves (early pair {sqrt, rsqrt}).
    HIPArray rsqrt_of_sqrt_map_(Value numerator) const {
        HIPArray r;
        if constexpr (IsFloat) {
            const auto *d = m_buf->deferred;
            int exponent = 0;
            if (!detail::hip_defer_gather_flag() || d->index_type != EK_SQRT || d->scaled || !(numerator == numerator) ||
                numerator - numerator != Value(0) || numerator == Value(0))
                return r;
            const Value mant = std::frexp(numerator, &exponent);
            if ((mant != Value(0.5) && mant != Value(-0.5)) || exponent < -32 || exponent > 32) return r;
            detail::HIPBuffer *src = d->table;
            auto *nd = new typename detail::HIPBuffer::Deferred{ src, nullptr, nullptr, Type, EK_RSQRT, sizeof(Value), false, 1, nullptr };
            nd->scaled = numerator != Value(1);
            nd->scale_bits = imm_bits(numerator);
            src->ref_count++;
            r.m_buf = new detail::HIPBuffer();
            r.m_buf->size = m_buf->size;
            r.m_buf->deferred = nd;
            r.m_buf->pending_link();
            src->readers.push_back(r.m_buf);
        }
        return r;
    }
    /// The products that the derivatives of rcp and rsqrt are made of (autodiff.h:381-403: -sqr(result), -.5 result sqr(result)),
    /// taken of UNEVALUATED maps of one source: rcp(u) rcp(u), rsqrt(u) rsqrt(u), rsqrt(u) (rsqrt(u) rsqrt(u)) stay unevaluated as
    /// ONE map of u each (EK_RCP_SQR, EK_RSQRT_SQR, EK_RSQRT_CUBE: the same roundings as the eager products), so that the
    /// weight remains a function of u the bucket-ordered consumers can form themselves.  Invalid array: not such a product.
    static HIPArray product_of_maps_(const HIPArray &a, const HIPArray &b) {
        HIPArray r;
        if constexpr (IsFloat) {
            if (!detail::hip_defer_gather_flag() || !a.mapped_() || !b.mapped_()) return r;
            const auto *da = a.m_buf->deferred, *db = b.m_buf->deferred;
            if (da->table != db->table || da->scaled || db->scaled) return r;
            const int oa = da->index_type, ob = db->index_type;
            int op = -1;
            if (a.m_buf == b.m_buf && oa == EK_RCP) op = EK_RCP_SQR;
            else if (a.m_buf == b.m_buf && oa == EK_RSQRT) op = EK_RSQRT_SQR;
            else if ((oa == EK_RSQRT && ob == EK_RSQRT_SQR) || (oa == EK_RSQRT_SQR && ob == EK_RSQRT)) op = EK_RSQRT_CUBE;
            if (op < 0) return r;
            detail::HIPBuffer *src = da->table;
            auto *nd = new typename detail::HIPBuffer::Deferred{ src, nullptr, nullptr, Type, op, sizeof(Value), false, 1, nullptr };
            src->ref_count++;
            r.m_buf = new detail::HIPBuffer();
            r.m_buf->size = a.m_buf->size;
            r.m_buf->deferred = nd;
            r.m_buf->pending_link();
            src->readers.push_back(r.m_buf);
        }
        return r;
    }
    Value map_scale_() const {
        Value v = Value(1);
        if constexpr (IsFloat) {
            if (mapped_() && m_buf->deferred->scaled) memcpy(&v, &m_buf->deferred->scale_bits, sizeof(Value));
        }
        return v;
    }

    /// op(gather(A, idx), x, gather(C, idx)) (fma family, shared index array, no mask, x an array) stays unevaluated as a
    /// kind-2 node when the library's bucket-ordered path covers the shape (see detail::HIPBuffer::Deferred); returns an
    /// invalid array otherwise.  The two gathers are marked as consumed, exactly as if the fused kernel had run.
    static HIPArray defer_pair_fma_(int op, const HIPArray &ga, const HIPArray &x, const HIPArray &gc, size_t n) {
        HIPArray r;
        if constexpr (IsFloat) {
            const auto *p = ga.m_buf->deferred, *q = gc.m_buf->deferred;
            // (both gathers under the SAME mask array -- or none: masked-out lanes are dropped by the partition)
            const char *shape = "fma(gather(A, idx), x, gather(B, idx))";
            if (!detail::hip_defer_gather_flag()) return r;
            if (p->mask != q->mask) { detail::hip_note_element_order(shape, "the two gathers use different mask arrays"); return r; }
            if (x.m_is_imm || !x.m_buf || x.m_buf->size != n) { detail::hip_note_element_order(shape, "x is a scalar / of another length"); return r; }
            if (!x.m_buf->owned || x.m_buf->exported) { detail::hip_note_element_order(shape, "x is a view of foreign memory or has a zero-copy export (an external writer could change it)"); return r; }
            if (x.m_buf->deferred) { detail::hip_note_element_order(shape, "x is itself unevaluated"); return r; }
            if (p->mask && sizeof(Value) != 4) { detail::hip_note_element_order(shape, "masked gathers of 8-byte elements"); return r; }
            if (!ek_hip_bucketed_applicable(Type, p->index_type, p->table->size, n)) {
                detail::hip_note_element_order(shape, "the library does not cover the shape (fewer than 2^18 lookups, a table within one bucket or beyond "
                                                      "64 slices, deterministic mode, ENOKI_HIP_BUCKET_ORDERED=0)");
                return r;
            }
            auto *d = new typename detail::HIPBuffer::Deferred{ p->table, p->index, p->mask, Type, p->index_type, sizeof(Value),
                                                                false, 2, nullptr };
            d->table2 = q->table;
            d->arg0 = x.m_buf;
            d->op = op;
            r.m_buf = new detail::HIPBuffer();
            r.m_buf->size = n;
            r.m_buf->deferred = d;
            r.m_buf->pending_link();
            detail::HIPBuffer *seen[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
            int k = 0;
            for (detail::HIPBuffer *src : { d->table, d->index, d->table2, d->arg0, d->mask }) {
                if (!src) continue;
                src->ref_count++;
                bool dup = false;
                for (int j = 0; j < k; ++j) dup = dup || seen[j] == src;
                if (!dup) { src->readers.push_back(r.m_buf); seen[k++] = src; }
            }
            ga.m_buf->deferred->consumed = true;
            gc.m_buf->deferred->consumed = true;
        }
        return r;
    }

    /// Horizontal reduction over map_op(u) of an unevaluated kind-2 node `u` in bucket order; false: not covered (the caller
    /// evaluates u in element order).  `held_by_consumer`: references on u that belong to the array being reduced.
    static bool reduce_bucketed_(detail::HIPBuffer *u, int op, int map_op, void *out, uint32_t held_by_consumer, int keep_op = EK_COPY) {
        if constexpr (!IsFloat) {
            return false;
        } else {
            // the sum of one half of an unevaluated sincos pair whose other half is still held: the shape of a derivative that
            // the tape will ask for (see ek_hip_bucketed_pair_create_hinted)
            if (map_op != EK_COPY && !map_on_load_(map_op)) return false;          // (second-wave maps: element order, before a partition is made for nothing)
            const bool adjoint_expected = op == EK_HSUM && keep_op != EK_COPY && ek_hip_bucketed_early_pair(map_op, keep_op);
            // (sin / cos: both functions bounded by 1 -- the adjoint sums can then be formed in 64-bit fixed point, see enoki_hip.h)
            const bool bounded = (map_op == EK_SIN || map_op == EK_COS) && (keep_op == EK_SIN || keep_op == EK_COS);
            ek_hip_bucketed *b = u->bucketed(adjoint_expected ? (unsigned) EK_BUCKETED_HINT_ADJOINT | (bounded ? (unsigned) EK_BUCKETED_HINT_BOUNDED : 0u) : 0u);
            if (!b) return false;
            // somebody else can still ask for u (the cos(u) of the derivative, a user handle): keep it in bucket order
            const int keep = u->ref_count > held_by_consumer ? 1 : 0;
            int rc = ek_hip_bucketed_reduce(b, op, map_op, out, keep, keep_op);
            if (rc == EK_ERR_UNSUPPORTED) return false;
            detail::hip_check(rc, "HIPArray (bucket-ordered reduction)");
            return true;
        }
    }

    static constexpr size_t gather_multi_small_ = (size_t) 3 << 20, gather_multi_large_ = (size_t) 128 << 20;

    /// gather_multi_ restricted to the shapes that go through staged records (used by DiffArray, whose other struct
    /// gathers stay one deferred gather per component so that their consumers can fuse them)
    template <size_t N, typename Index>
    static bool gather_records_(const HIPArray *sources, HIPArray *results, const Index &index, const MaskType &mask) {
        if constexpr (IsMask || (sizeof(Value) != 4 && sizeof(Value) != 8) || N < 2 || N > 4) {
            return false;
        } else {
            for (size_t c = 0; c < N; ++c)
                if (sources[c].size() <= 1 || sources[c].size() != sources[0].size()) return false;
            if (ek_hip_gather_multi_plan(Type, Index::Type, (int) N, sources[0].size(),
                                         broadcast_size(index.size(), mask.size())) != EK_GATHER_RECORDS)
                return false;
            return gather_multi_<N>(sources, results, index, mask);
        }
    }

    /// N tables, one index / mask array: a single kernel that reads the indices once (Array<HIPArray, N> sources)
    template <size_t N, typename Index>
    static bool gather_multi_(const HIPArray *sources, HIPArray *results, const Index &index, const MaskType &mask) {
        if constexpr (IsMask || (sizeof(Value) != 4 && sizeof(Value) != 8) || N < 2 || N > 4) {
            return false;
        } else {
            size_t table_bytes = 0;
            bool same_size = true;
            for (size_t c = 0; c < N; ++c) {
                if (sources[c].size() <= 1) return false;          // broadcast components take the generic path
                table_bytes = std::max(table_bytes, sources[c].size() * sizeof(Value));
                same_size &= sources[c].size() == sources[0].size();
            }
            // One launch reads the indices once, but its working set is ALL tables: when one table fits the 4 MiB L2 of
            // an XCD and the set does not, separate launches are up to 2x faster (profiles/probe_gather_multi_r01.txt)
            static const int policy = [] {
                const char *e = getenv("ENOKI_HIP_GATHER_MULTI");
                return !e ? 0 : (e[0] == 'a' ? 1 : e[0] == 'n' ? 2 : 0);      // always / never / (default) by size
            }();
            // components of one length (the usual structure of arrays): the library picks between staged {x, y, ..}
            // records, one launch over all tables and one launch per table (ek_hip_gather_multi_sized)
            const bool sized = policy == 0 && same_size &&
                ek_hip_gather_multi_plan(Type, Index::Type, (int) N, sources[0].size(),
                                         broadcast_size(index.size(), mask.size())) == EK_GATHER_RECORDS;
            if (policy == 2 || (policy == 0 && !sized && N * table_bytes > gather_multi_small_ && table_bytes < gather_multi_large_))
                return false;
            size_t n = broadcast_size(index.size(), mask.size());
            void *outs[N];
            const void *bases[N];
            for (size_t c = 0; c < N; ++c) {
                results[c] = empty_(n);
                outs[c] = results[c].m_buf->ptr;
                bases[c] = sources[c].data();
            }
            ek_operand oi = index.operand(), om = mask.operand();
            if (sized)
                detail::hip_check(ek_hip_gather_multi_sized(Type, Index::Type, (int) N, outs, bases, sources[0].size(), &oi, &om, n),
                                  "gather_multi_");
            else
                detail::hip_check(ek_hip_gather_multi(Type, Index::Type, (int) N, outs, bases, &oi, &om, n), "gather_multi_");
            return true;
        }
    }

    /// What a user-level scatter / scatter_add does with a target that other handles share.  Default: copy on write -- arrays
    /// are values, the other handles keep the old contents.  With hip_set_scatter_aliasing(true) (ENOKI_HIP_SCATTER_ALIASING=1):
    /// write IN PLACE, every handle observes the scatter -- what copies of a CUDAArray do, which alias one JIT variable
    /// (cuda.h:224-226; cuda_var_mark_dirty).  Unevaluated nodes that read the buffer are evaluated first either way (they
    /// were created before the write), and the Tape's own sweeps always copy on write (they share buffers as values).
    void make_writable_() {
        auto &sh = detail::HIPBuffer::shared();
        if (sh.scatter_alias && sh.sweeps == 0 && m_buf && !m_is_imm) {
            m_buf->force_readers();
            m_buf->leave_narrowed_cache();
            m_buf->drop_host_mirror();
            ptr_();
        } else {
            make_unique();
        }
    }
    /// Tape sweeps bracket themselves with this (autodiff_impl.h): inside, scatters copy on write whatever the mode
    static void sweep_scope_(bool enter) { detail::HIPBuffer::shared().sweeps += enter ? 1 : -1; }

    template <bool IsPermute, typename Index>
    static void scatter_array_(HIPArray &target, const HIPArray &value, const Index &index, const MaskType &mask) {
        target.make_writable_();
        value.template scatter_<sizeof(Value)>(target.data(), detach(index), mask);
    }

    template <bool IsPermute, typename Index>
    static void scatter_add_array_(HIPArray &target, const HIPArray &value, const Index &index, const MaskType &mask) {
        target.make_writable_();
        value.template scatter_add_<sizeof(Value)>(target.data(), detach(index), mask, target.size());
    }

    /// `count` scatter_adds through ONE index / mask array into `count` arrays of equal size; weights[c] (may be null) is
    /// multiplied onto values[c] inside the kernel with safe_mul semantics.  Issued by Tape::backward() for gathers that
    /// share their index array: the indices are read and binned once (ek_hip_scatter_add_multi).
    template <typename Index>
    static void scatter_add_multi_(size_t count, HIPArray *const *targets, const HIPArray *const *values,
                                   const HIPArray *const *weights, const Index &index, const MaskType &mask) {
        constexpr size_t kMax = 4;
        if (count == 0 || count > kMax) throw std::runtime_error("HIPArray::scatter_add_multi_(): 1 to 4 streams expected");
        index.require_valid("scatter_add_multi_"); mask.require_valid("scatter_add_multi_");
        size_t n = broadcast_size(index.size(), mask.size());
        if constexpr (IsFloat) {
            if (scatter_add_bucketed_(count, targets, values, weights, index, mask)) return;
        }
        void *bases[kMax];
        ek_operand ov[kMax], ow[kMax];
        const ek_operand *pv[kMax], *pw[kMax];
        int ops[kMax];
        bool any_weight = false, any_map = false;
        for (size_t c = 0; c < count; ++c) {
            values[c]->require_valid("scatter_add_multi_");
            if (targets[c]->size() != targets[0]->size())
                throw std::runtime_error("HIPArray::scatter_add_multi_(): the targets must have the same size");
            n = broadcast_size(n, values[c]->size());
            ops[c] = EK_COPY;
            bool in_place = false;
            if constexpr (IsFloat) {
                // an unevaluated unary result (the cos(u) of d/du sin(u), ...) is applied while the stream is loaded --
                // unless a target is its own source buffer
                if (values[c]->mapped_() && !values[c]->m_buf->deferred->scaled && map_on_load_(values[c]->m_buf->deferred->index_type)) {
                    const auto *d = values[c]->m_buf->deferred;
                    in_place = true;
                    for (size_t t = 0; t < count; ++t) {
                        in_place = in_place && targets[t]->m_buf != d->table;
                        // the same array as a WEIGHT is evaluated below (operand()), which would release this node
                        in_place = in_place && !(weights && weights[t] && weights[t]->m_buf == values[c]->m_buf);
                    }
                    if (in_place) {
                        if (d->table->deferred) d->table->force();      // the map of an unevaluated fma of gathers
                        ov[c] = ek_operand{ d->table->ptr, 0, d->table->size };
                        ops[c] = d->index_type;
                        any_map = true;
                    }
                }
            }
            if (!in_place) ov[c] = values[c]->operand();
            pv[c] = &ov[c];
            pw[c] = nullptr;
            if (weights && weights[c]) {
                weights[c]->require_valid("scatter_add_multi_");
                n = broadcast_size(n, weights[c]->size());
                ow[c] = weights[c]->operand();
                pw[c] = &ow[c];
                any_weight = true;
            }
            targets[c]->make_unique();
            bases[c] = targets[c]->data();
        }
        ek_operand oi = index.operand(), om = mask.operand();
        if (any_map)
            detail::hip_check(ek_hip_scatter_add_multi_map(Type, Index::Type, (int) count, bases, targets[0]->size(), pv, ops,
                                                           any_weight ? pw : nullptr, &oi, &om, n, 0), "scatter_add_multi_");
        else
            detail::hip_check(ek_hip_scatter_add_multi(Type, Index::Type, (int) count, bases, targets[0]->size(), pv,
                                                       any_weight ? pw : nullptr, &oi, &om, n, 0), "scatter_add_multi_");
    }

    /// The adjoint of gathers whose results went into ONE unevaluated fma `u = op(gather(A, idx), x, gather(C, idx))`: when every
    /// value stream is a unary function of that u (or a host scalar), every weight is u's own x, and index is u's index array,
    /// the streams are evaluated and accumulated bucket by bucket on u's partition -- no count / scan / partition of the
    /// indices in the backward sweep.  false: not this shape (the caller takes the element-order pipeline).
    template <typename Index>
    static bool scatter_add_bucketed_(size_t count, HIPArray *const *targets, const HIPArray *const *values,
                                      const HIPArray *const *weights, const Index &index, const MaskType &mask) {
        if ((mask.m_is_imm && !mask.m_imm) || !index.m_buf || index.m_is_imm) return false;
        detail::HIPBuffer *mask_buf = mask.m_is_imm ? nullptr : mask.m_buf;      // must be u's own mask (checked below)
        detail::HIPBuffer *u = nullptr;
        int from_u[4], ops[4], weighted[4];
        uint64_t imm[4], scale[4];
        for (size_t c = 0; c < count; ++c) {
            const HIPArray &v = *values[c];
            detail::HIPBuffer *src = nullptr;
            from_u[c] = 1; ops[c] = EK_COPY; imm[c] = 0;
            scale[c] = imm_bits(v.map_scale_());
            if (v.mapped_() && !map_on_load_(v.m_buf->deferred->index_type)) return false;      // (a second-wave map: the bucket-ordered kernels do not carry it)
            if (v.mapped_()) { src = v.m_buf->deferred->table; ops[c] = v.m_buf->deferred->index_type; }
            else if (v.paired_()) src = v.m_buf;
            else if (v.m_is_imm) { from_u[c] = 0; imm[c] = imm_bits(v.m_imm); }
            else return false;
            if (src) {
                if (!src->deferred || src->deferred->kind != 2 || (u && u != src)) return false;
                u = src;
            }
            weighted[c] = weights && weights[c] ? 1 : 0;
            if (weighted[c] && weights[c]->m_is_imm) {
                // a host-scalar weight (the -1 that `p - gather(B, idx)` records for its subtrahend): the stream's scale takes it
                // when the product has the bits of safe_mul(w, scale * f(u)) -- one of the two factors is +-1
                Value sc, w = weights[c]->m_imm;
                memcpy(&sc, &scale[c], sizeof(Value));
                if (w == Value(0) || !(w == w) || (sc != Value(1) && sc != Value(-1) && w != Value(1) && w != Value(-1))) return false;
                scale[c] = imm_bits(sc * w);
                weighted[c] = 0;
            }
        }
        if (!u) {
            // only host scalars: the partition still pays when a weight names the x of a pending fma over this index array
            for (size_t c = 0; c < count && !u; ++c)
                if (weighted[c] && weights[c]->m_buf)
                    for (detail::HIPBuffer *rd : weights[c]->m_buf->readers)
                        if (rd->deferred && rd->deferred->kind == 2 && rd->deferred->arg0 == weights[c]->m_buf &&
                            rd->deferred->index == index.m_buf && rd->deferred->bucketed) { u = rd; break; }
            if (!u) return false;
        }
        const auto *d = u->deferred;
        if (d->index != index.m_buf || d->index_type != Index::Type || u->size != index.m_buf->size || d->mask != mask_buf) return false;
        for (size_t c = 0; c < count; ++c) {
            if (weighted[c] && (weights[c]->m_is_imm || weights[c]->m_buf != d->arg0)) return false;
            if (targets[c]->size() != d->table->size) return false;
        }
        // writers first: a target that aliases one of u's sources evaluates u (and drops its partition) right here.  A target
        // that is a pending zeros node held by nobody else (the fresh gradient buffer of the sweep) is not memset: the fold
        // WRITES its sums there.
        void *bases[4];
        int fresh[4];
        for (size_t c = 0; c < count; ++c) {
            fresh[c] = targets[c]->zeroed_() && targets[c]->m_buf->ref_count == 1 ? 1 : 0;
            for (size_t t = 0; t < c; ++t) fresh[c] = fresh[c] && targets[t]->m_buf != targets[c]->m_buf;
            if (!fresh[c]) targets[c]->make_unique();
        }
        if (!u->deferred) return false;
        ek_hip_bucketed *b = u->bucketed();
        if (!b) return false;
        for (size_t c = 0; c < count; ++c) {
            if (fresh[c]) targets[c]->m_buf->adopt_uninitialized();
            bases[c] = targets[c]->m_buf->ptr;
        }
        int rc = ek_hip_bucketed_scatter_add_scaled(b, (int) count, bases, from_u, ops, imm, weighted, fresh, scale);
        if (rc != EK_OK) {
            // not covered after all: the adopted targets become what they promised to be before anybody else adds to them
            for (size_t c = 0; c < count; ++c)
                if (fresh[c]) detail::hip_check(ek_hip_memset(bases[c], 0, targets[c]->m_buf->size * sizeof(Value)), "scatter_add_multi_");
            if (rc == EK_ERR_UNSUPPORTED) return false;
            detail::hip_check(rc, "scatter_add_multi_ (bucket order)");
        }
        return true;
    }

    /// An external consumer received this buffer's address (see make_unique())
    void mark_exported_() const {
        if (!m_buf) return;
        m_buf->force_readers();                // unevaluated nodes that read this buffer must see it as it is NOW
        m_buf->exported = true;
        // an external view may write: neither a narrowed copy OF this buffer nor this buffer AS somebody's narrowed copy stays valid
        m_buf->drop_narrowed();
        m_buf->leave_narrowed_cache();
    }

    /// Same device buffer (or the same host-known scalar)?  Lets the tape recognise gathers that share an index array.
    bool same_storage_(const HIPArray &o) const {
        if (m_is_imm || o.m_is_imm) return m_is_imm && o.m_is_imm && imm_bits(m_imm) == imm_bits(o.m_imm);
        return m_buf && m_buf == o.m_buf;
    }

    // -----------------------------------------------------------------------------------------
    //  Horizontal operations (cuda.h:693-794)
    // -----------------------------------------------------------------------------------------

    HIPArray hsum_() const { return reduce(EK_HSUM, "hsum_"); }
    HIPArray hprod_() const { return reduce(EK_HPROD, "hprod_"); }
    HIPArray hmin_() const { return reduce(EK_HMIN, "hmin_"); }
    HIPArray hmax_() const { return reduce(EK_HMAX, "hmax_"); }

    bool all_() const { return mask_reduce(EK_ALL, "all_") != 0; }
    bool any_() const { return mask_reduce(EK_ANY, "any_") != 0; }
    size_t count_() const { return (size_t) mask_reduce(EK_COUNT, "count_"); }

    /// Stream compaction (cuda.h:907-923 / horiz.cu:124-160): keeps the entries whose mask is set, in order.
    /// Built from the library's own scan + scatter; reads the new size back (synchronises, like the reference).
    HIPArray compress_(const MaskType &mask) const {
        if (mask.size() == 0) return HIPArray();
        if (size() == 1) return *this;                       // broadcast operand: returned as is (cuda.h:910-911)
        if (mask.size() != size()) throw std::runtime_error("HIPArray::compress_(): size mismatch!");
        using UInt32 = HIPArray<uint32_t>;
        UInt32 ones = UInt32::select_(mask, UInt32(1u), UInt32(0u));
        UInt32 pos = ones.psum_();                         // inclusive prefix sum: 1-based slot of kept entries
        size_t kept = (size_t) pos.coeff(size() - 1);
        HIPArray result = empty_(kept);
        if (kept) scatter_<sizeof(Value)>(result.data(), pos.sub_(UInt32(1u)), mask);
        return result;
    }

    HIPArray reverse_() const {
        size_t n = size();
        if (n <= 1) return *this;
        HIPArray r = empty_(n);
        detail::hip_check(ek_hip_reverse(Type, r.m_buf->ptr, ptr_(), n), "reverse_");
        return r;
    }

    HIPArray psum_() const {
        size_t n = size();
        if (n <= 1) return *this;
        HIPArray r = empty_(n);
        detail::hip_check(ek_hip_psum(Type, r.m_buf->ptr, ptr_(), n), "psum_");
        return r;
    }

    // -----------------------------------------------------------------------------------------
    //  Storage access
    // -----------------------------------------------------------------------------------------

    size_t size() const { return m_is_imm ? 1 : (m_buf ? m_buf->size : 0); }
    size_t slices_() const { return size(); }
    bool empty() const { return size() == 0; }

    /// What state the array is in and what its consumers can still do with it (python: hip_explain(array))
    std::string explain_() const {
        if (m_is_imm) return "host scalar (passed to kernels as an argument)";
        if (!m_buf) return "uninitialized";
        const std::string n = std::to_string(m_buf->size) + " elements";
        if (!m_buf->deferred) return "evaluated array, " + n + (m_buf->exported ? ", exported zero-copy" : "") + (m_buf->owned ? "" : ", foreign memory");
        const auto *d = m_buf->deferred;
        switch (d->kind) {
            case 0: return "unevaluated gather from a table of " + std::to_string(d->table->size) + " entries, " + n +
                           (d->consumed ? " (a fused consumer has read it: the next access runs the gather kernel)"
                                        : ": add / sub / mul / fma consume it in place; fma (or `g * x + g2`) with a second gather through the SAME index "
                                          "array stays unevaluated for bucket order");
            case 1: return std::string("unevaluated unary op ") + std::to_string(d->index_type) + (d->scaled ? " times a host scalar" : "") + ", " + n +
                           ": reductions and scatter_add value streams apply it while loading; source " +
                           (d->table->deferred ? "unevaluated (kind " + std::to_string(d->table->deferred->kind) + ")" : "evaluated");
            case 2: return "unevaluated " + std::string(!d->table2 ? "product of a gather with an array" : d->op >= EK_MULADD ? "product-then-sum of two gathers through one index array" : "fma of two gathers through one index array") + " (K = " +
                           std::to_string(d->table->size) + ", " + n + "): BUCKET ORDER possible -- a horizontal reduction (directly or through one fusable "
                           "unary op) and the adjoint scatter_add of the gathers stay in bucket order; any other access evaluates it in element order" +
                           (d->bucketed ? "; partition built" : "");
            case 3: return "zeros that nobody has looked at, " + n;
            case 4: return "unevaluated arithmetic op " + std::to_string(d->op) + " of arity " + std::to_string(d->arity) + " over evaluated operands, " + n +
                           ": a reduction (through up to three fusable unary ops) reads the operands once; any other access runs the kernel";
            default: return "unevaluated (kind " + std::to_string(d->kind) + ")";
        }
    }

    /// True when the array is a host-known scalar equal to `v` (lets callers skip `x * 1` style passes)
    bool is_literal_(Value v) const { return m_is_imm && m_imm == v; }
    bool valid() const { return m_is_imm || m_buf != nullptr; }
    bool is_immediate() const { return m_is_imm; }

    /// Device pointer; an immediate is materialised on first use
    const Value *data() const { materialize(); return m_buf ? (const Value *) ptr_() : nullptr; }
    Value *data() {
        materialize();
        if (!m_buf) return nullptr;
        m_buf->force_readers();                // the caller may write through the pointer
        m_buf->leave_narrowed_cache();
        m_buf->drop_host_mirror();
        return (Value *) ptr_();
    }

    /// Host-side iteration `for (float v : array)` (the reference's CUDAArray iterates its managed memory, cuda.h:945-949):
    /// a read-only host copy made on first use (synchronises); discarded when a mutable pointer is handed out
    const Value *begin() const {
        if (!valid() || size() == 0) return nullptr;
        materialize();
        if (!m_buf->host_mirror) {
            void *h = malloc(m_buf->size * sizeof(Value));
            if (!h) throw std::bad_alloc();
            if (ek_hip_memcpy_to_host(h, ptr_(), m_buf->size * sizeof(Value)) != EK_OK) {
                free(h);
                detail::hip_raise("HIPArray::begin");
            }
            m_buf->host_mirror = h;
        }
        return (const Value *) m_buf->host_mirror;
    }
    const Value *end() const { const Value *b = begin(); return b ? b + size() : nullptr; }

    /// Broadcast a size-1 array / set the size of an empty array (CUDAArray::resize, cuda.h:935-937)
    void resize(size_t size) { set_slices_(size); }

    void set_slices_(size_t new_size) {
        size_t cur = size();
        if (cur == new_size) return;
        if (cur == 0) {
            release();
            allocate(new_size);
        } else if (cur == 1) {
            HIPArray r = empty_(new_size);
            ek_operand oa = operand();
            detail::hip_check(ek_hip_unary(EK_COPY, Type, r.m_buf->ptr, &oa, new_size), "set_slices");
            *this = std::move(r);
        } else {
            throw std::runtime_error("HIPArray::resize(): only arrays of size 0 or 1 can be resized (have " +
                                     std::to_string(cur) + ", requested " + std::to_string(new_size) + ")");
        }
    }

    /// Fetch one element (synchronises; cuda.h:939-943)
    Value coeff(size_t i) const {
        if (m_is_imm) return m_imm;
        if (!m_buf || i >= m_buf->size)
            throw std::runtime_error("HIPArray::coeff(): index " + std::to_string(i) + " out of range");
        std::conditional_t<IsMask, uint8_t, Value> v;
        detail::hip_check(ek_hip_memcpy_to_host(&v, (const uint8_t *) ptr_() + i * sizeof(Value), sizeof(Value)), "coeff");
        return (Value) v;
    }
    Value operator[](size_t i) const { return coeff(i); }
    /// x[mask] = value  (masked assignment proxy, array_base.h:144-157)
    auto operator[](const MaskType &mask) { return masked(*this, mask); }

    /// Copy everything to the host (synchronises)
    std::vector<std::conditional_t<IsMask, uint8_t, Value>> to_host() const {
        std::vector<std::conditional_t<IsMask, uint8_t, Value>> out(size());
        if (m_is_imm) out[0] = m_imm;
        else if (!out.empty())
            detail::hip_check(ek_hip_memcpy_to_host(out.data(), ptr_(), out.size() * sizeof(Value)), "to_host");
        return out;
    }

    /// First entry whose mask bit is set (array_router.h extract(); synchronises).  Undefined for an empty mask.
    Value extract_(const MaskType &mask) const {
        if (size() <= 1) return coeff(0);
        using UInt32 = HIPArray<uint32_t>;
        UInt32 lane = UInt32::arange_(0, (ptrdiff_t) size(), 1);
        uint32_t first = UInt32::select_(mask, lane, UInt32(~0u)).hmin_().coeff(0);
        if (first == ~0u) throw std::runtime_error("extract_(): the mask is empty");
        return coeff(first);
    }

    /// One PCG32 draw as a single fused kernel (enoki/random.h; reference random.h:68-133): advances
    /// `state` where `mask` is set and returns the sample derived from the old state.  `kind` is an
    /// ek_pcg32_kind that must match Value (u32 / f32 / u64 / f64).
    static HIPArray pcg32_next_(int kind, HIPArray<uint64_t> &state, const HIPArray<uint64_t> &inc, const MaskType &mask) {
        size_t n = broadcast_size(broadcast_size(state.size(), inc.size()), mask.size());
        if (n == 0) throw std::runtime_error("pcg32_next_(): uninitialized generator");
        HIPArray result = empty_(n);
        HIPArray<uint64_t> next = HIPArray<uint64_t>::empty_(n);
        ek_operand os = state.operand(), oi = inc.operand(), om = mask.operand();
        detail::hip_check(ek_hip_pcg32_next(kind, result.m_buf->ptr, (uint64_t *) next.m_buf->ptr, &os, &oi, &om, n),
                          "pcg32_next_");
        state = std::move(next);
        return result;
    }

    /// No-ops of the eager backend that keep templated code written for the JIT backend compiling
    HIPArray &eval() { return *this; }
    const HIPArray &eval() const { return *this; }
    HIPArray &managed() { return *this; }
    const HIPArray &managed() const { return *this; }

    ek_operand operand() const {
        ek_operand o;
        if (m_is_imm) {
            o.ptr = nullptr;
            o.imm = imm_bits(m_imm);
            o.size = 1;
        } else {
            o.ptr = m_buf ? ptr_() : nullptr;
            o.imm = 0;
            o.size = m_buf ? m_buf->size : 0;
        }
        return o;
    }

    void require_valid(const char *what) const {
        if (!valid())
            throw std::runtime_error(std::string("HIPArray::") + what + "(): operand is uninitialized");
    }

    /// Writers (scatter targets) must not alias other handles
    void make_unique() {
        materialize();
        if (!m_buf) return;
        m_buf->force_readers();                // deferred gathers from this array see its contents before the write
        m_buf->leave_narrowed_cache();         // (a cache entry holds a reference of its own: without this the copy below would
                                               //  always be taken, with it a sole user handle writes in place)
        m_buf->drop_host_mirror();
        if (m_buf->ref_count > 1) {
            // copy on write.  An exported buffer is parked (one reference is never given back): the external view keeps
            // reading valid -- if from now on stale -- memory after the other handles are gone
            if (m_buf->exported) { m_buf->exported = false; m_buf->ref_count++; }
            HIPArray r = empty_(m_buf->size);
            detail::hip_check(ek_hip_memcpy_device(r.m_buf->ptr, ptr_(), m_buf->size * sizeof(Value)), "make_unique");
            *this = std::move(r);
        } else {
            ptr_();
        }
    }

    static size_t broadcast_size(size_t a, size_t b) {
        if (a == b || b == 1) return a;
        if (a == 1) return b;
        throw std::runtime_error("HIPArray: arrays of incompatible size (" + std::to_string(a) + " and " +
                                 std::to_string(b) + ")");
    }

private:
    static uint64_t imm_bits(Value v) {
        uint64_t bits = 0;
        if constexpr (IsMask) bits = v ? 1 : 0;
        else memcpy(&bits, &v, sizeof(Value));
        return bits;
    }

    void allocate(size_t size) {
        m_buf = new detail::HIPBuffer();
        m_buf->size = size;
        m_is_imm = false;
        if (ek_hip_malloc((size ? size : 1) * sizeof(Value), &m_buf->ptr) != EK_OK) {
            delete m_buf;
            m_buf = nullptr;
            detail::hip_raise("HIPArray::allocate");
        }
    }

    void release() {
        if (m_buf && --m_buf->ref_count == 0) delete m_buf;
        m_buf = nullptr;
    }
public:
    /// An expiring array (the temporary argument of a routed unary function, array.h) lets go of its buffer early
    void release_expiring_() { release(); }
private:

    void materialize() const {
        if (!m_is_imm) return;
        HIPArray *self = const_cast<HIPArray *>(this);
        Value v = m_imm;
        self->allocate(1);
        detail::hip_check(ek_hip_fill(Type, self->m_buf->ptr, imm_bits(v), 1), "materialize");
    }

    HIPArray unary(int op, const char *what) const {
        require_valid(what);
        if constexpr (!IsMask) {
            if (m_is_imm && op == EK_NEG) {
                if constexpr (IsFloat) return HIPArray(Value(-m_imm));
                else return HIPArray((Value) (std::make_unsigned_t<Value>(0) - (std::make_unsigned_t<Value>) m_imm));
            }
        }
        if constexpr (IsFloat) {
            if (op == EK_NEG && mapped_())
                if (HIPArray r = scaled_map_(Value(-1)); r.valid()) return r;
            if (map_fusable_(op) && can_defer_map_()) return defer_map_(op);
        }
        size_t n = size();
        HIPArray r = empty_(n);
        ek_operand oa = operand();
        detail::hip_check(ek_hip_unary(op, Type, r.m_buf->ptr, &oa, n), what);
        return r;
    }

    /// Device pointer of a valid buffer; a deferred gather / map is executed first
    void *ptr_() const {
        if (m_buf->deferred) m_buf->force();
        return m_buf->ptr;
    }

    /// add / sub / mul / fma with a deferred gather among the operands: the gather is consumed in place
    /// (ek_hip_map_gathered).  Returns false when the combination is not fusable; operand() then materialises.
    static bool map_gathered_(int arity, int op, const HIPArray *const *x, size_t n, HIPArray &result, const char *what) {
        if constexpr (!IsFloat) {
            return false;
        } else {
            bool d[3] = { false, false, false };
            for (int k = 0; k < arity; ++k) d[k] = x[k]->deferred_() && x[k]->m_buf->size == n;
            if (d[0] && d[1]) d[1] = false;                                  // one gathered factor per product
            if (arity == 3 && d[2] && (d[0] || d[1])) {
                // a gathered factor AND a gathered addend: one 8-byte lookup when they share index, mask and table size
                const auto *p = x[d[0] ? 0 : 1]->m_buf->deferred, *q = x[2]->m_buf->deferred;
                const bool shared = p->index == q->index && p->mask == q->mask && p->index_type == q->index_type &&
                                    p->table->size == q->table->size;
                // the parameter lookup of BASELINE config 3b: left unevaluated for a consumer that may take it bucket by bucket
                if (shared) {
                    if (HIPArray r = defer_pair_fma_(op, *x[d[0] ? 0 : 1], *x[d[0] ? 1 : 0], *x[2], n); r.valid()) {
                        result = std::move(r);
                        return true;
                    }
                }
                if (!shared) detail::hip_note_element_order("fma(gather(A, i), x, gather(B, j))", "the gathers do not share index array, mask and table size");
                if (!shared || p->table->size * 8 > n) d[2] = false;     // interleaving K records has to pay for itself
            }
            // the same array in a fused AND an unfused slot (g * g): the unfused use materialises it anyway
            for (int k = 0; k < arity; ++k)
                for (int j = 0; j < arity; ++j)
                    if (!d[k] && d[j] && x[k]->m_buf == x[j]->m_buf) d[j] = false;
            if (!d[0] && !d[1] && !d[2]) return false;
            ek_gathered g[3];
            ek_operand o[3];
            const ek_gathered *pg[3] = { nullptr, nullptr, nullptr };
            const ek_operand *po[3] = { nullptr, nullptr, nullptr };
            for (int k = 0; k < arity; ++k) {
                if (d[k]) { g[k] = x[k]->m_buf->gathered(); pg[k] = &g[k]; }
                else { o[k] = x[k]->operand(); po[k] = &o[k]; }
            }
            result = empty_(n);
            int rc = ek_hip_map_gathered(arity, op, Type, result.m_buf->ptr, po, pg, n);
            if (rc == EK_ERR_UNSUPPORTED) return false;
            detail::hip_check(rc, what);
            for (int k = 0; k < arity; ++k)
                if (d[k] && x[k]->m_buf->deferred) x[k]->m_buf->deferred->consumed = true;
            return true;
        }
    }

    HIPArray binary(int op, const HIPArray &b, const char *what) const {
        require_valid(what); b.require_valid(what);
        if (m_is_imm && b.m_is_imm) {
            Value r;
            if (detail::host_binary<Value>(op, m_imm, b.m_imm, r)) return HIPArray(r);
        }
        if constexpr (IsFloat) {
            // an unevaluated map times a host scalar stays an unevaluated map (safe_mul(w, c) with c != 0 is w * c up to the
            // sign of a zero, like the unit-weight shortcuts of the tape; ENOKI_HIP_STRICT_SAFE_MUL=1 evaluates literally)
            if (op == EK_MUL || (op == EK_SAFE_MUL && !detail::hip_strict_safe_mul())) {
                const HIPArray *m = mapped_() && b.m_is_imm ? this : (b.mapped_() && m_is_imm ? &b : nullptr);
                const HIPArray *c = m == this ? &b : this;
                if (m && c->m_imm != Value(0))
                    if (HIPArray r = m->scaled_map_(c->m_imm); r.valid()) return r;
            }
            if (op == EK_MUL && mapped_() && b.mapped_())
                if (HIPArray r = product_of_maps_(*this, b); r.valid()) return r;
            // a power of two over an unevaluated sqrt(u) is an unevaluated multiple of rsqrt(u) (the derivative's factor of sqrt)
            if (op == EK_DIV && m_is_imm && b.mapped_())
                if (HIPArray r = b.rsqrt_of_sqrt_map_(m_imm); r.valid()) return r;
        }
        size_t n = broadcast_size(size(), b.size());
        if constexpr (IsFloat) {
            // an unevaluated map times an evaluated array (the sweep's safe_mul(x, grad_u) with grad_u = cos(u) still pending):
            // the map and the product are the two outputs of one pass over the chain's operands
            if ((op == EK_MUL || op == EK_SAFE_MUL) && mapped_() != b.mapped_()) {
                const HIPArray &m = mapped_() ? *this : b, &w = mapped_() ? b : *this;
                const auto *d = m.m_buf->deferred;
                if (!w.m_is_imm && w.m_buf && !w.m_buf->deferred && w.m_buf->size == m.m_buf->size && n == m.m_buf->size && n > 1 &&
                    !(d->partner && d->partner->deferred)) {
                    HIPArray r = empty_(n);
                    m.m_buf->force_map_product(w.m_buf, op, r.m_buf->ptr);
                    return r;
                }
            }
            // `gather(A, idx) * x + gather(B, idx)` written with operators (BASELINE.json spells config 3b `a*x+b`): the product
            // stays unevaluated (kind 2 without an addend), the sum with the second gather makes the node of the whole expression
            if (op == EK_MUL && deferred_() != b.deferred_()) {
                const HIPArray &g = deferred_() ? *this : b, &x = deferred_() ? b : *this;
                if (HIPArray r = defer_gathered_product_(g, x, n); r.valid()) return r;
            }
            if ((op == EK_ADD || op == EK_SUB) && ((gathered_product_() && b.deferred_()) || (b.gathered_product_() && deferred_()))) {
                HIPArray r = gathered_product_() ? gathered_product_plus_(op == EK_ADD ? EK_MULADD : EK_MULSUB, b, n)
                                                 : b.gathered_product_plus_(op == EK_ADD ? EK_MULADD : EK_NMULADD, *this, n);
                if (r.valid()) return r;
            }
        }
        if ((op == EK_ADD || op == EK_SUB || op == EK_MUL) && (deferred_() || b.deferred_())) {
            const HIPArray *x[3] = { this, &b, nullptr };
            HIPArray r;
            if (map_gathered_(2, op, x, n, r, what)) return r;
        }
        if constexpr (IsFloat) {
            // `a * x + b` written with operators: the product stays unevaluated (kind 4), the sum that consumes it becomes the
            // product-then-sum node -- two roundings, like the eager kernels, in one pass whenever a reduction consumes it
            if ((op == EK_ADD || op == EK_SUB) && (arith_() || b.arith_())) {
                HIPArray r;
                if (arith_() && !b.arith_()) r = product_plus_(op == EK_ADD ? EK_MULADD : EK_MULSUB, b, n);
                else if (b.arith_() && !arith_()) r = b.product_plus_(op == EK_ADD ? EK_MULADD : EK_NMULADD, *this, n);
                if (r.valid()) return r;
            }
            if (op == EK_MUL && !mapped_() && !b.mapped_()) {
                const HIPArray *x[3] = { this, &b, nullptr };
                if (HIPArray r = defer_arith_(2, EK_MUL, x, n); r.valid()) return r;
            }
        }
        HIPArray r = empty_(n);
        ek_operand oa = operand(), ob = b.operand();
        detail::hip_check(ek_hip_binary(op, Type, r.m_buf->ptr, &oa, &ob, n), what);
        return r;
    }

    HIPArray ternary(int op, const HIPArray &b, const HIPArray &c, const char *what) const {
        require_valid(what); b.require_valid(what); c.require_valid(what);
        size_t n = broadcast_size(broadcast_size(size(), b.size()), c.size());
        if (op != EK_SAFE_FMADD && (deferred_() || b.deferred_() || c.deferred_())) {
            const HIPArray *x[3] = { this, &b, &c };
            HIPArray g;
            if (map_gathered_(3, op, x, n, g, what)) return g;
        }
        if constexpr (IsFloat) {
            if (op == EK_FMADD || op == EK_FMSUB || op == EK_FNMADD || op == EK_FNMSUB) {
                const HIPArray *x[3] = { this, &b, &c };
                if (HIPArray r = defer_arith_(3, op, x, n); r.valid()) return r;
            }
        }
        HIPArray r = empty_(n);
        ek_operand oa = operand(), ob = b.operand(), oc = c.operand();
        detail::hip_check(ek_hip_ternary(op, Type, r.m_buf->ptr, &oa, &ob, &oc, n), what);
        return r;
    }

    MaskType compare(int op, const HIPArray &b, const char *what) const {
        require_valid(what); b.require_valid(what);
        size_t n = broadcast_size(size(), b.size());
        MaskType r = MaskType::empty_(n);
        ek_operand oa = operand(), ob = b.operand();
        detail::hip_check(ek_hip_compare(op, Type, (uint8_t *) r.data(), &oa, &ob, n), what);
        return r;
    }

    HIPArray reduce(int op, const char *what) const {
        size_t n = size();
        if (n == 1) return *this;
        HIPArray r = empty_(1);
        if constexpr (IsFloat) {
            if (mapped_()) {
                const auto *d = m_buf->deferred;
                detail::HIPBuffer *src = d->table;
                // map(u) with u an unevaluated fma of gathers: gathers, fma, map and reduction bucket by bucket
                // scale * f(u): sums are linear -- reduce f(u) as usual, scale the result (the order of the roundings is that of
                // a reduction anyway); other reductions of a scaled map see the evaluated array
                const bool scaled = d->scaled;
                if (!scaled || op == EK_HSUM) {
                    auto finish = [&]() -> HIPArray { return scaled ? r.binary(EK_MUL, HIPArray(map_scale_()), what) : r; };
                    if (src->deferred && src->deferred->kind == 2) {
                        // Which function of u will be asked for next?  Another unevaluated map of the same u that somebody holds
                        // is the derivative's factor (the cos(u) next to sin(u), -sin(u) next to cos(u), rcp(u) next to log(u));
                        // a map that is itself held elsewhere is its own (exp).  THAT is kept in bucket order rather than u --
                        // or, for y = hsum(f(u)), summed per table entry on the spot (ek_hip_bucketed_pair_create_hinted).
                        int keep_op = EK_COPY, others = 0;
                        for (detail::HIPBuffer *rd : src->readers)
                            if (rd != m_buf && rd->deferred && rd->deferred->kind == 1 && rd->deferred->table == src) {
                                keep_op = rd->deferred->index_type;
                                ++others;
                            }
                        if (others == 0 && m_buf->ref_count > 1) keep_op = d->index_type;
                        else if (others != 1) keep_op = EK_COPY;
                        // (a value that somebody else holds as well will be asked for again: u counts as held, too)
                        if (reduce_bucketed_(src, op, d->index_type, r.m_buf->ptr, m_buf->ref_count > 1 ? 0 : 1, keep_op)) return finish();
                    }
                    if (src->deferred && (src->deferred->kind == 1 || src->deferred->kind == 4)) {
                        // maps over maps over an unevaluated arithmetic node: whatever only this chain wants is applied on load
                        ek_chain ch;
                        m_buf->build_chain(ch, /* peek = */ true);
                        if (ch.n_maps > 1 || ch.arity > 1) {
                            detail::hip_check(ek_hip_reduce_chain(op, Type, r.m_buf->ptr, &ch, n), what);
                            return finish();
                        }
                    }
                    if (src->deferred) src->force();
                    detail::hip_check(ek_hip_reduce_map(op, d->index_type, Type, r.m_buf->ptr, src->ptr, n), what);
                    return finish();
                }
            }
            if (paired_() && reduce_bucketed_(m_buf, op, EK_COPY, r.m_buf->ptr, 1)) return r;
            if (arith_() && m_buf->ref_count == 1 && m_buf->readers.empty()) {
                // hsum(fmadd(a, x, b)) and the like: the operands are read once, the result of the arithmetic is never written
                const auto *a = m_buf->deferred;
                ek_chain ch;
                ch.arity = a->arity; ch.base_op = a->op; ch.n_maps = 0;
                for (int k = 0; k < 3; ++k) {
                    ch.map_ops[k] = EK_COPY;
                    const detail::HIPBuffer *b = k < a->arity ? a->operand_buf(k) : nullptr;
                    ch.src[k] = k >= a->arity ? ek_operand{ nullptr, 0, 0 }
                              : a->is_imm[k]  ? ek_operand{ nullptr, a->imm[k], 1 } : ek_operand{ b->ptr, 0, b->size };
                }
                detail::hip_check(ek_hip_reduce_chain(op, Type, r.m_buf->ptr, &ch, n), what);
                return r;
            }
        }
        detail::hip_check(ek_hip_reduce(op, Type, r.m_buf->ptr, m_buf ? ptr_() : nullptr, n), what);
        return r;
    }

    uint64_t mask_reduce(int op, const char *what) const {
        static_assert(IsMask, "all_/any_/count_ require a mask array");
        if (m_is_imm) return m_imm ? 1 : 0;
        uint64_t result = 0;
        detail::hip_check(ek_hip_mask_reduce(op, m_buf ? (const uint8_t *) ptr_() : nullptr, size(), &result), what);
        return result;
    }

    detail::HIPBuffer *m_buf = nullptr;
    Value m_imm = Value(0);
    bool m_is_imm = false;
};

/// Runtime helpers mirroring cuda_eval / cuda_sync / cuda_whos / cuda_malloc_trim (cuda.h:109-200)
inline void hip_eval() { }
inline void hip_sync() { detail::hip_check(ek_hip_sync(), "hip_sync"); }

/// Sharding across the GPUs of a node from C++, one process per GPU (ek_hip_dist_*, include/enoki_hip.h: RCCL on the library
/// stream).  `hip_dist_unique_id` on rank 0, ship the 128 bytes, `hip_dist_init` everywhere; arrays are index-range shards
/// (`hip_dist_shard_range`), tables and scalars replicated, so vertical operations are local and only horizontal results
/// (`hsum_all`, ...) and the gradients of replicated tables (`all_reduce_`) cross ranks.
inline void hip_dist_unique_id(void *id128) { detail::hip_check(ek_hip_dist_unique_id(id128), "hip_dist_unique_id"); }
inline void hip_dist_init(int rank, int world, const void *id128) { detail::hip_check(ek_hip_dist_init(rank, world, id128), "hip_dist_init"); }
inline void hip_dist_finalize() { detail::hip_check(ek_hip_dist_finalize(), "hip_dist_finalize"); }
inline std::pair<size_t, size_t> hip_dist_shard_range(size_t n, int rank, int world) {
    size_t b = 0, e = 0;
    detail::hip_check(ek_hip_dist_shard_range(n, rank, world, &b, &e), "hip_dist_shard_range");
    return { b, e };
}
/// sum / product / min / max over all ranks, entry by entry (reduce_op: EK_HSUM ...): the replicated result
template <typename T> HIPArray<T> all_reduce_(const HIPArray<T> &x, int reduce_op = EK_HSUM) {
    HIPArray<T> r = x;
    if (r.size() == 0) return r;
    r.make_unique();
    detail::hip_check(ek_hip_dist_all_reduce(HIPArray<T>::Type, reduce_op, r.data(), r.size()), "all_reduce_");
    return r;
}
/// hsum over the shards of all ranks (local two-stage reduction, then a 1-element all-reduce)
template <typename T> HIPArray<T> hsum_all(const HIPArray<T> &shard) { return all_reduce_(hsum(shard), EK_HSUM); }

/// Step graphs (ek_hip_graph_*, include/enoki_hip.h).  Use these wrappers rather than the C entry points: arrays that are still
/// unevaluated when a capture starts (deferred gathers / unary results) are evaluated first, so that none of them receives
/// its storage from the graph's private pool.
inline void hip_graph_begin() {
    detail::HIPBuffer::force_all_pending();
    detail::hip_check(ek_hip_graph_begin(), "hip_graph_begin");
}
inline ek_hip_graph *hip_graph_end() {
    // arrays that are still unevaluated get their kernels and their storage INSIDE the graph (while the stream is still
    // capturing): every replay then refreshes them like any other array that is alive at the end of the capture
    try {
        detail::HIPBuffer::force_all_pending();
    } catch (...) {
        ek_hip_graph *dead = nullptr;
        if (ek_hip_graph_end(&dead) == EK_OK && dead) ek_hip_graph_destroy(dead);
        throw;
    }
    ek_hip_graph *g = nullptr;
    detail::hip_check(ek_hip_graph_end(&g), "hip_graph_end");
    return g;
}
inline void hip_graph_launch(ek_hip_graph *g) { detail::hip_check(ek_hip_graph_launch(g), "hip_graph_launch"); }
inline void hip_graph_destroy(ek_hip_graph *g) { detail::hip_check(ek_hip_graph_destroy(g), "hip_graph_destroy"); }
inline void hip_malloc_trim() { detail::hip_check(ek_hip_malloc_trim(), "hip_malloc_trim"); }
inline std::string hip_whos() {
    char *w = ek_hip_whos();
    std::string s(w ? w : "");
    free(w);
    return s;
}

/// Deferred evaluation of gathers and fusable unary ops on / off (on by default; ENOKI_HIP_DEFER=0 switches it off for a
/// whole process).  Off: every operation runs its own kernel when it is called.
inline void hip_set_defer(bool value) { detail::hip_defer_gather_flag() = value; }
inline bool hip_defer() { return detail::hip_defer_gather_flag(); }
inline void hip_set_defer_gather(bool value) { hip_set_defer(value); }
/// Shared-handle scatter semantics of the reference's CUDAArray (see HIPArray::make_writable_): off by default
inline void hip_set_scatter_aliasing(bool value) { detail::HIPBuffer::shared().scatter_alias = value; }
inline bool hip_scatter_aliasing() { return detail::HIPBuffer::shared().scatter_alias; }
inline bool hip_defer_gather() { return hip_defer(); }

template <typename T> inline void set_label(const HIPArray<T> &, const char *) { }

} // namespace enoki
