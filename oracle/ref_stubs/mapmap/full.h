// Stand-in for mapMAP's mapmap/full.h (an un-vendored download).  NOT a solver: the classes below only RECORD the model
// the reference's view_selection.cpp builds through them (nodes, edges in insertion order, per-node label sets, unary
// costs, the Potts weight, the termination criterion, the control block), and mapMAP::optimize hands that model to a
// hook the test installs, which answers with one label OFFSET per node (the oracle's solver runs behind it).  What
// oracle/_ref pins through this is the reference's model construction and its decode (label_from_offset, the
// "Incorrect labeling" guard, label 0 for unseen faces) -- not an optimiser: parity with mapMAP's own output stays
// unpinned.  Test infrastructure only.
#ifndef MVS_REF_STUB_MAPMAP_FULL_H
#define MVS_REF_STUB_MAPMAP_FULL_H
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <vector>
namespace mapmap {
typedef std::uint64_t luint_t;
template <typename C> constexpr unsigned int sys_max_simd_width() { return 1; }
template <typename C, unsigned int W> using _iv_st = std::int32_t;
template <typename C, unsigned int W> using _s_t = C;

struct Model {
    std::uint64_t n_nodes;
    std::vector<std::uint32_t> edges;          // 2 per edge, insertion order
    std::vector<float> edge_weights;
    bool components_updated;
    std::vector<std::vector<std::int32_t> > labels;
    bool compress;
    std::vector<std::vector<float> > costs;
    std::vector<int> unary_set;                 // set_unary(i, &unaries[i]) seen, and for the right node
    float potts;
    int term_window; double term_ratio;
    int ctrl[9]; std::uint64_t seed;
};
inline Model& model() { static Model m; return m; }
struct SolveHook { int (*fn)(void* user); void* user; std::vector<std::int32_t> offsets; };
inline SolveHook& solve_hook() { static SolveHook h; return h; }

template <typename C>
class Graph {
public:
    explicit Graph(std::size_t n) { Model& m = model(); m = Model(); m.n_nodes = n; }
    void add_edge(std::size_t a, std::size_t b, C w) { Model& m = model(); m.edges.push_back((std::uint32_t)a); m.edges.push_back((std::uint32_t)b); m.edge_weights.push_back(w); }
    void update_components() { model().components_updated = true; }
};
template <typename C, unsigned int W>
class LabelSet {
public:
    LabelSet(std::size_t n, bool compress) { model().labels.assign(n, std::vector<std::int32_t>()); model().compress = compress; }
    void set_label_set_for_node(std::size_t i, std::vector<_iv_st<C, W> > const& l) { model().labels.at(i) = l; }
    _iv_st<C, W> label_from_offset(std::size_t i, _iv_st<C, W> offset) const { return model().labels.at(i).at((std::size_t)offset); }
};
template <typename C, unsigned int W>
class UnaryTable {
public:
    UnaryTable(std::size_t node, LabelSet<C, W>*) : node(node) { if (model().costs.size() <= node) model().costs.resize(node + 1); }
    void set_costs(std::vector<_s_t<C, W> > const& c) { model().costs.at(node) = c; }
    std::size_t node;
};
template <typename C, unsigned int W>
class PairwisePotts { public: explicit PairwisePotts(C w) : w(w) {} C w; };
template <typename C, unsigned int W>
class StopWhenReturnsDiminish { public: StopWhenReturnsDiminish(int window, double ratio) : window(window), ratio(ratio) {} int window; double ratio; };
enum TREE_SAMPLER_ALGORITHM { OPTIMISTIC_TREE_SAMPLER, LOCK_FREE_TREE_SAMPLER };
struct mapMAP_control {
    bool use_multilevel, use_spanning_tree, use_acyclic, force_acyclic, relax_acyclic_maximal, sample_deterministic;
    int spanning_tree_multilevel_after_n_iterations, min_acyclic_iterations;
    TREE_SAMPLER_ALGORITHM tree_algorithm;
    std::uint64_t initial_seed;
};
template <typename C, unsigned int W>
class mapMAP {
public:
    void set_graph(Graph<C>*) {}
    void set_label_set(LabelSet<C, W>*) {}
    void set_unary(std::size_t i, UnaryTable<C, W>* u) { Model& m = model(); if (m.unary_set.size() <= i) m.unary_set.resize(i + 1, 0); m.unary_set[i] = (u && u->node == i) ? 1 : -1; }
    void set_pairwise(PairwisePotts<C, W>* p) { model().potts = p->w; }
    void set_logging_callback(std::function<void(const luint_t, const _iv_st<C, W>)> cb) { log = cb; }
    void set_termination_criterion(StopWhenReturnsDiminish<C, W>* t) { model().term_window = t->window; model().term_ratio = t->ratio; }
    void optimize(std::vector<_iv_st<C, W> >& solution, mapMAP_control const& c) {
        Model& m = model();
        const int ctrl[9] = {c.use_multilevel, c.use_spanning_tree, c.use_acyclic, c.spanning_tree_multilevel_after_n_iterations, c.force_acyclic,
                             c.min_acyclic_iterations, c.relax_acyclic_maximal, (int)c.tree_algorithm, c.sample_deterministic};
        for (int i = 0; i < 9; ++i) m.ctrl[i] = ctrl[i];
        m.seed = c.initial_seed;
        SolveHook& h = solve_hook();
        if (!h.fn) throw std::runtime_error("oracle/_ref: no solver hook installed");
        h.offsets.assign(m.n_nodes, 0);
        if (h.fn(h.user) != 0) throw std::runtime_error("oracle/_ref: solver hook failed");
        solution.assign(h.offsets.begin(), h.offsets.end());
        if (log) log(0, 0);
    }
private:
    std::function<void(const luint_t, const _iv_st<C, W>)> log;
};
}  // namespace mapmap
#endif
