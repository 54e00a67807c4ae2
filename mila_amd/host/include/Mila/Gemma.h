// Gemma-4 decoder on the CDNA4 device: GemmaConfig, and GemmaTransformer<TWeightQuant> with
//   * decode()/prefill() in the REFERENCE's per-component order (one C-ABI call per component
//     forward, exactly the sequence of GemmaBlock::decode / ::prefill,
//     /root/reference/Mila/Src/Dnn/Components/Transformers/Gemma/Gemma.Block.ixx:197-356, and
//     GemmaTransformer::decode, Gemma.ixx:281-297), and
//   * decodeFused(): the same arithmetic as 6 launches per layer, and
//   * a hipGraph of the fused step with the position in device memory (one replay per token).
// The three paths produce bit-identical logits (tests/test_gemma_host_gpu.py).
//
// Quirks kept from the reference (SURVEY.md Appendix A "Gemma block quirks"): sandwich norms; per-head
// q/k/v norms; RoPE after q/k norm and never on V; global layers have no V projection, V =
// v_norm(raw k_proj); residuals use the block input and res1; output scaled by a per-layer scalar;
// embedding scaled by sqrt(D) with a second bf16 rounding; layer i is global iff (i+1) % 6 == 0;
// tied table: bf16 for NoWeightQuant, per-row FP8 for quantized bodies (Gemma.ixx:143-147).
#pragma once
#include <functional>
#include <map>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>

#include "GemmaBlock.h"
#include "Serialization.h"

namespace Mila::Dnn
{
    struct GemmaConfig
    {
        // defaults = Gemma-4 12B (Gemma.Config.ixx:796-821)
        dim_t vocab_size = 262144, embedding_dim = 3840, num_layers = 48, num_heads = 16, num_kv_heads = 8, head_dim = 256,
              hidden_dim = 15360, global_head_dim = 512, num_global_kv_heads = 1, window = 1024, sliding_window_pattern = 6,
              global_rotary_dim = 128;
        float rms_norm_eps = 1e-6f, rope_theta_local = 10000.0f, rope_theta_global = 1000000.0f, final_logit_softcapping = 30.0f;
        bool bounded_local_kv = false;   ///< SlidingWindowKvCache on the sliding-window layers (ring of window + chunk - 1 rows)

        bool isGlobalLayer( dim_t i ) const { return ( i + 1 ) % sliding_window_pattern == 0; }   // Gemma.Config.ixx:504-507
        dim_t headDim( bool g ) const { return g ? global_head_dim : head_dim; }
        dim_t numKvHeads( bool g ) const { return g ? num_global_kv_heads : num_kv_heads; }
        dim_t qWidth( bool g ) const { return num_heads * headDim( g ); }
        dim_t kvWidth( bool g ) const { return numKvHeads( g ) * headDim( g ); }
        dim_t packedQkvWidth( bool g ) const { return qWidth( g ) + ( g ? 1 : 2 ) * kvWidth( g ); }   // K=V on global layers
        dim_t windowFor( bool g ) const { return g ? 0 : window; }
        float embeddingScale() const { return std::sqrt( static_cast<float>( embedding_dim ) ); }

        void validate() const
        {
            if ( embedding_dim <= 0 || num_layers <= 0 || num_heads <= 0 || hidden_dim <= 0 || vocab_size <= 0 )
                throw std::invalid_argument( "GemmaConfig: dimensions must be positive" );
            if ( num_heads % num_kv_heads != 0 || num_heads % num_global_kv_heads != 0 )
                throw std::invalid_argument( "GemmaConfig: num_heads must be a multiple of the KV head counts" );
            if ( embedding_dim % 128 != 0 || hidden_dim % 128 != 0 || qWidth( false ) % 128 != 0 || qWidth( true ) % 128 != 0 )
                throw std::invalid_argument( "GemmaConfig: feature widths must be multiples of 128 (FP4 group size)" );
        }

        /// weight parameters streamed per decode token (Linear body, tied table) -- SURVEY.md section 8d
        dim_t linearParamsPerLayer( bool g ) const
        {
            return embedding_dim * packedQkvWidth( g ) + qWidth( g ) * embedding_dim + embedding_dim * 2 * hidden_dim + hidden_dim * embedding_dim;
        }
    };

    inline void hipCheck( hipError_t e, const char* what )
    {
        if ( e != hipSuccess ) throw Compute::RocmException( std::string( what ) + ": " + hipGetErrorString( e ) );
    }

    template<WeightQuantPolicy TWeightQuant>
    class GemmaTransformer
    {
    public:
        static constexpr DeviceType kDevice = DeviceType::Rocm;
        static constexpr TensorDataType kPrecision = TensorDataType::BF16;
        using TensorType = Tensor<kPrecision, Compute::RocmDeviceMemoryResource>;
        using TokenTensor = Tensor<TensorDataType::INT32, Compute::RocmDeviceMemoryResource>;
        using LogitsTensor = Tensor<TensorDataType::FP32, Compute::RocmDeviceMemoryResource>;
        using LinearType = Linear<kDevice, kPrecision, TWeightQuant>;
        using TableQuantizationPolicy = std::conditional_t<TWeightQuant::kIsQuantized, Quant::Weight::PerChannelFp8<>, NoWeightQuant>;
        using LmHeadLinearType = Linear<kDevice, kPrecision, TableQuantizationPolicy>;
        using RmsNormType = RmsNorm<kDevice, kPrecision>;
        using TokenEmbeddingType = TokenEmbedding<kDevice, TensorDataType::INT32, kPrecision, TableQuantizationPolicy>;
        // Gemma.ixx:153-154: local blocks take the model's KV policy, global blocks attend to the whole context
        using LocalBlockType = GemmaBlock<kDevice, kPrecision, false, TWeightQuant, Quant::KvCache::NoKvCompression>;
        using BoundedLocalBlockType = GemmaBlock<kDevice, kPrecision, false, TWeightQuant, Quant::KvCache::SlidingWindowKvCache>;
        using GlobalBlockType = GemmaBlock<kDevice, kPrecision, true, TWeightQuant, Quant::KvCache::NoKvCompression>;
        static constexpr int kFmt = Quant::Weight::abiWeightFormat<TWeightQuant>();
        static constexpr int kTableFmt = Quant::Weight::abiWeightFormat<TableQuantizationPolicy>();

        /// a block behind its kind-independent face: the children (input_norm ... fc_down, rope, res_1/2, geglu), layer_scalar, the KV
        /// cache surface, and IDecoderLayer's prefill / decode (the reference-order path)
        using Layer = GemmaBlockBase<kDevice, kPrecision, TWeightQuant>;
        /// the model's blocks, indexed and iterated as Layer&
        class LayerList
        {
        public:
            using Store = std::vector<std::shared_ptr<Layer>>;
            struct Iterator
            {
                typename Store::const_iterator p;
                Layer& operator*() const { return **p; }
                Iterator& operator++() { ++p; return *this; }
                bool operator!=( const Iterator& o ) const { return p != o.p; }
            };
            size_t size() const noexcept { return v_.size(); }
            Layer& operator[]( size_t i ) const { return *v_[ i ]; }
            Iterator begin() const { return Iterator{ v_.begin() }; }
            Iterator end() const { return Iterator{ v_.end() }; }
            void push_back( std::shared_ptr<Layer> l ) { l->index = v_.size(); v_.push_back( std::move( l ) ); }
            const std::shared_ptr<Layer>& shared( size_t i ) const { return v_[ i ]; }
        private:
            Store v_;
        };

        GemmaTransformer( const GemmaConfig& cfg, dim_t max_seq, dim_t max_prefill, DeviceId device = Compute::Device::Rocm( 0 ) )
            : cfg_( cfg ), max_seq_( max_seq ), max_prefill_( std::max<dim_t>( max_prefill, 1 ) )
        {
            cfg_.validate();
            if ( max_seq <= 0 ) throw std::invalid_argument( "GemmaTransformer: max_seq must be positive" );
            owned_ctx_ = Compute::createExecutionContext( device );
            ctx_ = Compute::cast_context<kDevice>( owned_ctx_.get() );
            buildAll();
        }

        ~GemmaTransformer()
        {
            destroyGraph();
            for ( auto& e : ov_ev_ ) if ( e ) (void)hipEventDestroy( e );
            if ( ov_stream_ ) (void)hipStreamDestroy( ov_stream_ );
        }

        Compute::RocmExecutionContext* context() const noexcept { return ctx_; }
        const GemmaConfig& config() const noexcept { return cfg_; }
        LayerList& layers() noexcept { return layers_; }

        // ------------------------------------------------------------------------------------
        // synthetic parameters: counter-based uniform values generated on the
        // device; Linear weights U(-1/sqrt(K), 1/sqrt(K)); norm weights 1 + 0.1 U(-1,1); table U(-a,a)
        // ------------------------------------------------------------------------------------
        /// multipliers on the generator above.  The defaults are SURVEY.md's unit-scale random weights, under which a bf16 rounding
        /// difference grows ~1.4x per block; a CONDITIONED profile (small post-norm weights = small residual updates, q/k norm weights
        /// that keep scores O(1), a layer scalar != 1, a larger embedding scale) behaves like a trained model: differences stay at the
        /// rounding floor, so whole-model logits can be held to 1e-3 (tests/test_gemma_conditioned_gpu.py, tests/ref_gemma.py)
        struct SyntheticProfile { float linear_gain = 1.0f, qk_norm_center = 1.0f, post_norm_center = 1.0f, layer_scalar = 1.0f, table_gain = 1.0f; };

        void initSynthetic( uint64_t seed, const SyntheticProfile& pr = SyntheticProfile{} )
        {
            destroyGraph();   // layer scalars are baked into the captured launches
            // the staging buffer holds the largest matrix of the model: the table, fc_gate_up, or -- on geometries whose attention is wider than the stream -- a packed qkv
            const dim_t attn_w = cfg_.num_heads * std::max( cfg_.head_dim, cfg_.global_head_dim );
            const dim_t qkv_w = attn_w + 2 * std::max( cfg_.num_kv_heads * cfg_.head_dim, cfg_.num_global_kv_heads * cfg_.global_head_dim );
            const size_t max_elems = static_cast<size_t>( std::max( { cfg_.vocab_size * cfg_.embedding_dim, cfg_.embedding_dim * 2 * cfg_.hidden_dim, cfg_.embedding_dim * qkv_w } ) );
            TensorType staging( ctx_->getDeviceId(), shape_t{ static_cast<dim_t>( max_elems ) } );
            auto fillLinear = [&]( auto& lin, uint64_t s, float gain )
            {
                const dim_t N = lin.getConfig().getOutputFeatures(), K = lin.getConfig().getInputFeatures();
                fill( staging.data(), N * K, s, gain / std::sqrt( static_cast<float>( K ) ), 0.0f );
                lin.loadWeightFromDevice( staging.data() );
                ctx_->synchronize();
            };
            auto fillNorm = [&]( RmsNormType& n, uint64_t s, float center = 1.0f ) { fill( n.getWeight()->data(), n.getConfig().dim(), s, 0.1f * center, center ); };
            for ( size_t i = 0; i < layers_.size(); ++i )
            {
                auto& L = layers_[ i ];
                const uint64_t b = seed * 1000003ull + i * 64ull;
                fillLinear( *L.qkv_proj, b + 1, pr.linear_gain ); fillLinear( *L.o_proj, b + 2, pr.linear_gain ); fillLinear( *L.fc_gate_up, b + 3, pr.linear_gain ); fillLinear( *L.fc_down, b + 4, pr.linear_gain );
                fillNorm( *L.input_norm, b + 5 ); fillNorm( *L.q_norm, b + 6, pr.qk_norm_center ); fillNorm( *L.k_norm, b + 7, pr.qk_norm_center );
                fill( L.v_norm->getWeight()->data(), L.v_norm->getConfig().dim(), 0, 0.0f, 1.0f );   // unit weight (Gemma.Block.ixx:880-886)
                fillNorm( *L.post_attn_norm, b + 8, pr.post_norm_center ); fillNorm( *L.pre_ffn_norm, b + 9 ); fillNorm( *L.post_ffn_norm, b + 10, pr.post_norm_center );
                L.layer_scalar = pr.layer_scalar;
            }
            fillNorm( *final_norm_, seed * 1000003ull + 64ull * layers_.size() + 1 );
            fillLinear( *lm_head_, seed * 1000003ull + 64ull * layers_.size() + 2, pr.table_gain );   // the tied table
            ctx_->synchronize();
        }

        // ------------------------------------------------------------------------------------
        // reference-order decode: one component forward per line of GemmaBlock::decode
        // ------------------------------------------------------------------------------------
        LogitsTensor& decode( const TokenTensor& token, dim_t position )
        {
            Compute::TraceRange tr( "gemma.decode(reference order)" );
            checkPosition( position, 1 );
            kv_fill_ = position + 1;
            embed( token.data(), 1, *hidden_[ 0 ] );
            TensorType* x = hidden_[ 0 ].get();
            for ( auto& L : layers_ ) x = &L.decode( x->view( shape_t{ 1, 1, cfg_.embedding_dim } ), position );
            auto& normed = final_norm_->forward( x->view( shape_t{ 1, 1, cfg_.embedding_dim } ) );
            head( normed.data() );
            return *logits_;
        }

        // ------------------------------------------------------------------------------------
        // fused decode: embedding + 6 launches per layer + head
        // ------------------------------------------------------------------------------------
        LogitsTensor& decodeFused( const TokenTensor& token, dim_t position )
        {
            Compute::TraceRange tr( "gemma.decode(fused)" );
            checkPosition( position, 1 );
            kv_fill_ = position + 1;
            enqueueFusedStep( token.data(), static_cast<int>( position ), nullptr );
            return *logits_;
        }

        /// capture the fused step once; afterwards replayGraph() advances one token per call.
        /// The token is read from `token` (device) and the position from an internal device counter.
        /// Every device pointer the captured nodes hold is model-owned and fixed for the model's lifetime (the split-attention partials
        /// live in attn_partials_, never in the growable context scratch), so a later prefill or Linear::forward that grows the context
        /// scratch cannot leave the graph pointing at freed memory.  A second capture replaces the first (the old graph is destroyed).
        void captureGraph( const TokenTensor& token, dim_t start_position )
        {
            Compute::TraceRange tr( "gemma.captureGraph" );
            setDevicePosition( start_position );
            destroyGraph();
            hipStream_t s = reinterpret_cast<hipStream_t>( ctx_->getStream() );
            ctx_->synchronize();
            hipCheck( hipStreamBeginCapture( s, hipStreamCaptureModeThreadLocal ), "hipStreamBeginCapture" );
            hipGraph_t g = nullptr;
            try
            {
                // greedy sampler (feeds the next replay) + position bump + publication of the token.  The sampler's FIRST stage runs in the lm_head's epilogue (every
                // workgroup leaves its best logit and index in the sampler scratch), its final reduction does the other three: one launch behind the head, not three
                int sampler_partials = 0;
                captured_band_end_ = bandBucket( start_position );
                enqueueFusedStep( token.data(), static_cast<int>( captured_band_end_ ), pos_dev_->data(), sample_in_graph_ ? &sampler_partials : nullptr );
                if ( sample_in_graph_ )
                    Compute::rocmCheck( mila_cdna4_sample_argmax_final_advance( const_cast<TokenTensor&>( token ).data(), sample_scratch_->data(), sample_scratch_->sizeInBytes(), sampler_partials,
                                                                                pos_dev_->data(), token_ring_ ? token_seq_ : nullptr, token_ring_, token_ring_ ? token_ring_size_ : 0,
                                                                                ctx_->getStream() ) );
                else
                    Compute::rocmCheck( mila_cdna4_advance_position( pos_dev_->data(), ctx_->getStream() ) );
            }
            catch ( ... ) { (void)hipStreamEndCapture( s, &g ); if ( g ) (void)hipGraphDestroy( g ); throw; }
            hipCheck( hipStreamEndCapture( s, &g ), "hipStreamEndCapture" );
            graph_ = g;
            const hipError_t e = hipGraphInstantiate( &graph_exec_, graph_, nullptr, nullptr, 0 );
            if ( e != hipSuccess ) { graph_exec_ = nullptr; destroyGraph(); hipCheck( e, "hipGraphInstantiate" ); }
            captured_token_ = token.data();
            captured_sample_in_graph_ = sample_in_graph_;
            captured_ring_ = token_ring_;
        }
        /// capture on first use, and again whenever the captured graph no longer matches what a replay must do: another token
        /// buffer, or a different sampler setting (a graph captured without the sampler node never writes the next token)
        /// ... or a position outside the band bucket the attention launches were captured for (csrc/attention.hip: band_bucket -- 4096, 8192, 16384, ... keys: an unwindowed
        /// layer's split geometry and kernel form follow the live length bucket, not the cache capacity).  Cheap: callers invoke it before every replay.
        void ensureGraph( const TokenTensor& token, dim_t start_position )
        {
            if ( !graph_exec_ || captured_token_ != token.data() || captured_sample_in_graph_ != sample_in_graph_ || captured_ring_ != token_ring_ ||
                 start_position + 1 > captured_band_end_ || ( start_position + 1 <= captured_band_end_ / 2 && captured_band_end_ > 4096 ) )
                captureGraph( token, start_position );
        }
        /// the live-length bucket of a position (the rule of csrc/attention.hip: band_bucket)
        dim_t bandBucket( dim_t position ) const
        {
            dim_t b = 4096;
            while ( b < position + 1 && b < max_seq_ ) b <<= 1;
            return std::min( b, max_seq_ );
        }
        bool graphCaptured() const noexcept { return graph_exec_ != nullptr; }
        /// kernel nodes of the captured decode step (0 before a capture): the launches one token costs on the graph path
        size_t graphNodeCount() const
        {
            if ( !graph_ ) return 0;
            size_t n = 0;
            hipCheck( hipGraphGetNodes( graph_, nullptr, &n ), "hipGraphGetNodes" );
            return n;
        }
        bool graphSamples() const noexcept { return graph_exec_ != nullptr && captured_sample_in_graph_; }
        /// when set, every replay ends with the greedy sampler writing the next token into the token buffer the graph reads from:
        /// a closed autoregressive loop with no host round trip.  Takes effect at the next ensureGraph() / captureGraph().
        void setSampleInGraph( bool on ) { sample_in_graph_ = on; }
        /// a caller-owned token ring in host-visible memory + its device sequence counter (GemmaModel's decode-ahead loop): with the sampler in the graph, the
        /// step's last node publishes the sampled token there (mila_cdna4_advance_position_snapshot).  Takes effect at the next ensureGraph(); nullptr = off
        void setTokenRing( unsigned long long* ring, int size, unsigned long long* seq_dev )
        {
            if ( ring && ( size <= 0 || !seq_dev ) ) throw std::invalid_argument( "GemmaTransformer::setTokenRing: a ring needs a positive size and a sequence counter" );
            token_ring_ = ring; token_ring_size_ = ring ? size : 0; token_seq_ = ring ? seq_dev : nullptr;
        }
        void setDevicePosition( dim_t position )
        {
            checkPosition( position, 1 );
            const int32_t p = static_cast<int32_t>( position );
            Compute::rocmCheck( mila_cdna4_memcpy_h2d( pos_dev_->data(), &p, 4, ctx_->getStream() ) );
            ctx_->synchronize();
        }
        void replayGraph()
        {
            if ( !graph_exec_ ) throw std::runtime_error( "GemmaTransformer::replayGraph: captureGraph() first" );
            Compute::TraceRange tr( "gemma.decode(graph replay)" );
            hipCheck( hipGraphLaunch( graph_exec_, reinterpret_cast<hipStream_t>( ctx_->getStream() ) ), "hipGraphLaunch" );
        }
        LogitsTensor& logits() { return *logits_; }

        /// greedy device sampler: token <- argmax(logits); the token never leaves the device
        /// (GemmaModel::enqueueSampleNext, Models/GemmaModel.ixx:568)
        void sampleGreedy( TokenTensor& token_out )
        {
            Compute::TraceRange tr( "gemma.sample(greedy)" );
            Compute::rocmCheck( mila_cdna4_sample_argmax_fp32( logits_->data(), token_out.data(), (int)cfg_.vocab_size, sample_scratch_->data(),
                                                               sample_scratch_->sizeInBytes(), ctx_->getStream() ) );
        }

        /// SamplingParams (Components/Transformers/SamplingParams.ixx): temperature <= 0 is greedy; top_k 0 / top_p >= 1 disable
        /// the truncations.  The uniform r in [0, 1) is drawn by the caller, as in the reference (host RNG, GemmaModel.ixx:568).
        struct SamplingParams { float temperature = 1.0f; int top_k = 0; float top_p = 1.0f; };
        /// token <- a draw from softmax(softcap(logits) / temperature) restricted by top-k / top-p; the Gemma final logit
        /// softcap (cfg.final_logit_softcapping) is applied here, at the sampler (Gemma.ixx:30-32)
        void sampleStochastic( TokenTensor& token_out, const SamplingParams& sp, float r )
        {
            if ( sp.temperature <= 0.0f ) { sampleGreedy( token_out ); return; }
            Compute::TraceRange tr( "gemma.sample(stochastic)" );
            const size_t need = mila_cdna4_sample_stochastic_scratch_bytes( (int)cfg_.vocab_size );
            void* scratch = ctx_->getScratch( need );
            Compute::rocmCheck( mila_cdna4_sample_stochastic_fp32( logits_->data(), token_out.data(), (int)cfg_.vocab_size, cfg_.final_logit_softcapping, sp.temperature,
                                                                   sp.top_k, sp.top_p, r, scratch, need, ctx_->getStream() ) );
        }

        // ------------------------------------------------------------------------------------
        // prefill: whole prompt as one chunk (288 GB: no 12 GB-card chunking);
        // logits for the last position only (Gemma.ixx:269-276)
        // ------------------------------------------------------------------------------------
        LogitsTensor& prefill( const TokenTensor& tokens, dim_t T, dim_t position_offset = 0 )
        {
            if ( T <= 0 || T > max_prefill_ ) throw std::invalid_argument( "GemmaTransformer::prefill: chunk length out of range" );
            Compute::TraceRange tr( "gemma.prefill" );
            checkPosition( position_offset, T );
            kv_fill_ = position_offset + T;
            const dim_t D = cfg_.embedding_dim;
            embed( tokens.data(), static_cast<int>( T ), *pf_x_[ 0 ] );
            TensorType* x = pf_x_[ 0 ].get();
            int flip = 0;
            const bool fused = fused_prefill_ && fusedPrefillApplicable();
            if ( fused && prefill_overlap_ && overlapApplicable( T ) ) return prefillOverlapped( tokens, T, position_offset );
            for ( size_t i = 0; i < layers_.size(); ++i )
            {
                if ( !fused ) { x = &layers_[ i ].prefill( x->view( shape_t{ 1, T, D } ), position_offset ); continue; }   // GemmaBlock::prefill, one component per step
                TensorType* out = pf_x_[ 1 - flip ].get();
                blockPrefillFused( layers_[ i ], *x, i > 0, *out, i + 1 < layers_.size() ? &layers_[ i + 1 ] : nullptr, static_cast<int>( T ), static_cast<int>( position_offset ) );
                x = out;
                flip = 1 - flip;
            }
            auto last = x->slice( static_cast<size_t>( ( T - 1 ) * D ), shape_t{ 1, 1, D } );
            auto& normed = final_norm_->forward( last );
            head( normed.data() );
            return *logits_;
        }

        /// algorithmic bytes one decode token must read: weights + scales + norm weights + KV band
        double decodeBytesPerToken( dim_t context ) const
        {
            double b = 0;
            for ( auto& L : layers_ )
            {
                b += L.qkv_proj->getParameterBytes() + L.o_proj->getParameterBytes() + L.fc_gate_up->getParameterBytes() + L.fc_down->getParameterBytes();
                const dim_t band = L.global ? context : std::min<dim_t>( context, cfg_.window );
                b += 2.0 * band * cfg_.kvWidth( L.global ) * 2;
                b += 2.0 * ( 4 * cfg_.embedding_dim + 2 * cfg_.headDim( L.global ) );
            }
            b += lm_head_->getParameterBytes();
            return b;
        }
        double weightBytes() const
        {
            double b = lm_head_->getParameterBytes();
            for ( auto& L : layers_ ) b += L.qkv_proj->getParameterBytes() + L.o_proj->getParameterBytes() + L.fc_gate_up->getParameterBytes() + L.fc_down->getParameterBytes();
            return b;
        }
        /// GemmaTransformer::rewindKvCache (Gemma.ixx): every block drops positions >= `position`; false when any block refuses (a bounded ring
        /// that has already evicted what the rewind would need) -- the caller then prefills from 0, which overwrites positionally
        bool rewindKvCache( dim_t position, dim_t written )
        {
            if ( position < 0 || position > written ) return false;
            bool ok = true;
            for ( auto& L : layers_ ) ok = L.rewindKvCache( position, written ) && ok;
            if ( ok ) kv_fill_ = position;
            return ok;
        }
        /// the same against the fill this transformer has tracked itself (prefill / decode / decodeFused calls; graph replays advance on the device, so a
        /// caller that replays -- GemmaModel -- passes the fill explicitly): positions beyond the fill are rejected (Gemma.Cuda.cpp:373-383)
        bool rewindKvCache( dim_t position ) { return rewindKvCache( position, kv_fill_ ); }
        dim_t kvFill() const noexcept { return kv_fill_; }
        /// incremental prefill (prompt-prefix reuse): positions [0, offset) stay resident, tokens [offset, T) are prefilled at their positions, in chunks of
        /// the built prefill size; logits of the last position (Gemma.ixx prefillFrom; Gemma.Cuda.cpp:338-391)
        LogitsTensor& prefillFrom( const TokenTensor& tokens, dim_t T, dim_t offset )
        {
            if ( offset < 0 || offset >= T ) throw std::invalid_argument( "GemmaTransformer::prefillFrom: offset " + std::to_string( offset ) + " outside the prompt of " + std::to_string( T ) + " tokens" );
            LogitsTensor* out = nullptr;
            for ( dim_t p0 = offset; p0 < T; p0 += max_prefill_ )
            {
                const dim_t n = std::min( max_prefill_, T - p0 );
                auto chunk = tokens.slice( static_cast<size_t>( p0 ), shape_t{ 1, n } );
                out = &prefill( chunk, n, p0 );
            }
            return *out;
        }
        void resetKVCache() { for ( auto& L : layers_ ) L.resetKVCache(); }
        LmHeadLinearType& lmHead() { return *lm_head_; }
        RmsNormType& finalNorm() { return *final_norm_; }
        TokenEmbeddingType& tokenEmbedding() { return *temb_; }

        /// launch only the dominant kernel (fc_gate_up fused matvec) of layer `i` -- used by the bench to
        /// time that kernel with HIP events on the model stream
        void launchGateUp( size_t i ) { fusedGateUp( layers_[ i ] ); }
        double gateUpBytes( size_t i ) const { return static_cast<double>( layers_[ i ].fc_gate_up->getParameterBytes() ); }
        /// the dominant decode kernel of the schedule: the fc_gate_up fused matvec of layer i
        void launchDominant( size_t i )
        {
            if ( !cur_hidden_ ) cur_hidden_ = hidden_[ 0 ]->data();   // no fused step has run yet (reference-order timing)
            fusedGateUp( layers_[ i % layers_.size() ] );
        }
        double dominantBytes( size_t i ) const { return gateUpBytes( i % layers_.size() ); }

    private:
        void destroyGraph() noexcept
        {
            if ( graph_exec_ ) (void)hipGraphExecDestroy( graph_exec_ );
            if ( graph_ ) (void)hipGraphDestroy( graph_ );
            graph_exec_ = nullptr; graph_ = nullptr; captured_token_ = nullptr;
        }
        void fill( uint16_t* dst, dim_t n, uint64_t seed, float amp, float offset )
        {
            Compute::rocmCheck( mila_cdna4_fill_uniform_bf16( dst, n, seed, amp, offset, ctx_->getStream() ) );
        }

        template<typename C, typename... A> std::shared_ptr<C> make( const std::string& name, A&&... a )
        {
            auto c = std::make_shared<C>( name, std::forward<A>( a )... );
            c->setExecutionContext( ctx_ );
            return c;
        }

        void buildAll()
        {
            const dim_t D = cfg_.embedding_dim, P = max_prefill_;
            const auto dev = ctx_->getDeviceId();
            auto rms = [&]( dim_t dim ) { return RmsNormConfig( dim ).withEpsilon( cfg_.rms_norm_eps ).withBias( false ); };
            // workspace every block shares (they run one after the other): split scratch, attention / GeGLU / residual outputs, two stream buffers
            const dim_t maxq = std::max( cfg_.qWidth( false ), cfg_.qWidth( true ) ), maxkv = std::max( cfg_.kvWidth( false ), cfg_.kvWidth( true ) );
            q_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, maxq } );
            k_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, maxkv } );
            v_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, maxkv } );
            attn_out_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, maxq } );
            res1_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, D } );
            res2_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, D } );
            geglu_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, cfg_.hidden_dim } );
            for ( int i = 0; i < 2; ++i ) blk_out_[ i ] = std::make_shared<TensorType>( dev, shape_t{ 1, P, D } );
            // ... and one output buffer per component ROLE, shared by that role's component in every block (blocks run one after the other): the reference's pooled block
            // workspace (Gemma.ixx allocateBlockWorkspace: normed, qkv, q/k/v_normed, o, o_normed, ffn_in, gate_up, ffn_down, ffn_normed).  Per layer only the KV caches
            // and the parameters remain (Tests/.../Gemma.Cuda.cpp:494 StateMemory_PerLayerSlopeIsKvCacheNotActivations).
            const dim_t maxpacked = std::max( cfg_.packedQkvWidth( false ), cfg_.packedQkvWidth( true ) );
            ws_normed_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, D } );
            ws_o_normed_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, D } );
            ws_ffn_in_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, D } );
            ws_ffn_normed_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, D } );
            ws_q_normed_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, maxq } );
            ws_k_normed_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, maxkv } );
            ws_v_normed_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, maxkv } );
            ws_qkv_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, maxpacked } );
            ws_o_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, D } );
            ws_gate_up_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, 2 * cfg_.hidden_dim } );
            ws_down_ = std::make_shared<TensorType>( dev, shape_t{ 1, P, D } );
            for ( dim_t i = 0; i < cfg_.num_layers; ++i )
            {
                const std::string n = name_ + ".tf_layer_" + std::to_string( i );
                const bool g = cfg_.isGlobalLayer( i );
                GemmaBlockConfig bc;
                bc.model_dim = D; bc.hidden_dim = cfg_.hidden_dim; bc.num_heads = cfg_.num_heads; bc.num_kv_heads = cfg_.numKvHeads( g ); bc.head_dim = cfg_.headDim( g );
                bc.window = cfg_.windowFor( g ); bc.rotary_dim = g ? cfg_.global_rotary_dim : 0; bc.rope_theta = g ? cfg_.rope_theta_global : cfg_.rope_theta_local;
                bc.rms_norm_eps = cfg_.rms_norm_eps; bc.max_seq = max_seq_;
                // KV policy (Quantization/KvCache policies; CudaGqaOp.ixx:552-574): 288 GB of HBM makes unbounded caches the default;
                // SlidingWindowKvCache bounds the sliding-window layers to window + prefill_chunk - 1 rows, global layers stay unbounded
                auto wire = [&]( auto block )
                {
                    block->setExecutionContext( ctx_ );
                    block->installSharedWorkspace( q_, k_, v_, blk_out_[ i & 1 ] );
                    block->attn->installSharedOutput( attn_out_ );
                    block->geglu->installSharedOutput( geglu_ );
                    block->res_1->installSharedOutput( res1_ );
                    block->res_2->installSharedOutput( res2_ );
                    block->input_norm->installSharedOutput( ws_normed_ ); block->post_attn_norm->installSharedOutput( ws_o_normed_ );
                    block->pre_ffn_norm->installSharedOutput( ws_ffn_in_ ); block->post_ffn_norm->installSharedOutput( ws_ffn_normed_ );
                    block->q_norm->installSharedOutput( ws_q_normed_ ); block->k_norm->installSharedOutput( ws_k_normed_ ); block->v_norm->installSharedOutput( ws_v_normed_ );
                    block->qkv_proj->installSharedOutput( ws_qkv_ ); block->o_proj->installSharedOutput( ws_o_ );
                    block->fc_gate_up->installSharedOutput( ws_gate_up_ ); block->fc_down->installSharedOutput( ws_down_ );
                    block->build( BuildContext( shape_t{ 1, P, D }, RuntimeMode::Inference ) );
                    layers_.push_back( block );
                };
                if ( g ) wire( std::make_shared<GlobalBlockType>( n, bc ) );
                else if ( cfg_.bounded_local_kv ) wire( std::make_shared<BoundedLocalBlockType>( n, bc ) );
                else wire( std::make_shared<LocalBlockType>( n, bc ) );
            }
            // TokenEmbedding owns the raw table; the tied head adopts it before its build, so it never allocates its own (Gemma.ixx:659-684)
            temb_ = make<TokenEmbeddingType>( name_ + ".temb", TokenEmbeddingConfig().withVocabSize( cfg_.vocab_size ).withEmbeddingDim( D ).withEmbeddingScale( cfg_.embeddingScale() ) );
            temb_->build( BuildContext( shape_t{ 1, P }, RuntimeMode::Inference ) );
            final_norm_ = make<RmsNormType>( name_ + ".rmsn_final", rms( D ) );
            final_norm_->build( BuildContext( shape_t{ 1, 1, D }, RuntimeMode::Inference ) );
            lm_head_ = make<LmHeadLinearType>( name_ + ".lm_head", LinearConfig( D, cfg_.vocab_size ).withBias( false ) );
            if constexpr ( TableQuantizationPolicy::kIsQuantized ) lm_head_->installSharedWeight( temb_->getWeightTensorShared(), temb_->getWeightScalesTensorShared() );
            else lm_head_->installSharedWeight( temb_->getWeightTensorShared() );
            lm_head_->build( BuildContext( shape_t{ 1, 1, D }, RuntimeMode::Inference ) );

            for ( int i = 0; i < 3; ++i ) hidden_[ i ] = std::make_unique<TensorType>( dev, shape_t{ 1, 1, D } );
            for ( int i = 0; i < 2; ++i ) pf_x_[ i ] = std::make_unique<TensorType>( dev, shape_t{ 1, P, D } );
            pf_norm_ = std::make_unique<TensorType>( dev, shape_t{ 1, P, D } );
            pf_norm2_ = std::make_unique<TensorType>( dev, shape_t{ 1, P, D } );
            if constexpr ( kFmt != 0 )      // per-token e4m3 rows + scales of the fp8 x fp8 prefill paths (W4A8; W8A8 when the fp8 policy opts in)
                for ( int i = 0; i < 2; ++i )
                {
                    pf_q8_[ i ] = std::make_unique<TensorType>( dev, shape_t{ 1, P, ( D + 1 ) / 2 } );
                    pf_ts_[ i ] = std::make_unique<LogitsTensor>( dev, shape_t{ P } );
                }
            f_qkv_ = std::make_unique<TensorType>( dev, shape_t{ std::max( cfg_.packedQkvWidth( false ), cfg_.packedQkvWidth( true ) ) } );
            f_q_ = std::make_unique<TensorType>( dev, shape_t{ std::max( cfg_.qWidth( false ), cfg_.qWidth( true ) ) } );
            f_o_ = std::make_unique<TensorType>( dev, shape_t{ D } );
            f_down_ = std::make_unique<TensorType>( dev, shape_t{ D } );
            f_act_ = std::make_unique<TensorType>( dev, shape_t{ cfg_.hidden_dim } );
            logits_ = std::make_unique<LogitsTensor>( dev, shape_t{ 1, 1, cfg_.vocab_size } );
            pos_dev_ = std::make_unique<TokenTensor>( dev, shape_t{ 1 } );
            sample_scratch_ = std::make_unique<LogitsTensor>( dev, shape_t{ static_cast<dim_t>( mila_cdna4_sample_scratch_bytes() / 4 ) } );
            // split-attention partials of the fused / graph decode step: model-owned and never re-allocated (see captureGraph)
            attn_partials_ = std::make_unique<LogitsTensor>( dev, shape_t{ static_cast<dim_t>( ( attnScratchBytes() + 3 ) / 4 ) } );
            ctx_->synchronize();
        }

        void checkPosition( dim_t position, dim_t n ) const
        {
            if ( position < 0 || position + n > max_seq_ ) throw std::invalid_argument( "GemmaTransformer: position beyond the built sequence length" );
        }
        size_t attnScratchBytes() const
        {
            return std::max( mila_cdna4_attn_decode_scratch_bytes( 1, (int)cfg_.num_heads, (int)cfg_.head_dim ),
                             mila_cdna4_attn_decode_scratch_bytes( 1, (int)cfg_.num_heads, (int)cfg_.global_head_dim ) );
        }

        /// TokenEmbedding's gather (bf16 table, or FP8 table x row scale) and scale(sqrt(D)), token ids on the device
        void embed( const int32_t* tokens_dev, int n, TensorType& out ) { temb_->gather( tokens_dev, n, out ); }

        /// lm_head on the normalized last hidden state, fp32 logits (the 1e-3 bar is asserted on fp32)
        void head( const uint16_t* normed )
        {
            const float* sc = nullptr;
            if constexpr ( kTableFmt != 0 ) sc = lm_head_->getWeightScale()->data();
            Compute::rocmCheck( mila_cdna4_matvec_f32out( logits_->data(), normed, lm_head_->getWeight().rawData(), sc, kTableFmt, (int)cfg_.embedding_dim,
                                                          (int)cfg_.vocab_size, 0, ctx_->getStream() ) );
        }

        // ---- fused-glue prefill (the reference-order path is GemmaBlock::prefill) -----------------------
        /// act[T, F] = GeGLU(fc_gate_up(ffn_in)): one kernel when the fused GEMM serves the shape, else Linear + GeGLU
        /// `x8` / `ts` (W4A8 policy): ffn_in's rows already quantized per token by the tail that produced them -- the quantization launch is then skipped
        void gateUpGeglu( Layer& L, TensorType& ffn_in, TensorType& act, int T, TensorType* private_gate_up = nullptr, const uint8_t* x8_in = nullptr, const float* ts_in = nullptr )
        {
            const dim_t D = cfg_.embedding_dim;
            mila_stream_t st = ctx_->getStream();
            bool w4a8 = false;
            if constexpr ( kFmt == 2 )
                w4a8 = L.fc_gate_up->getOperation().fp8ActivationPrefill() && mila_cdna4_gemm_fp8_applicable( T, (int)D, 2 * (int)cfg_.hidden_dim );
            if constexpr ( kFmt == 1 )
            {
                // W8A8 (opt-in): the policy's e4m3 [2F, D] weights and per-channel scales straight into the fused fp8 x fp8 Linear + GeGLU kernel
                auto& op = L.fc_gate_up->getOperation();
                if ( op.fp8ActivationPrefill() && mila_cdna4_gemm_fp8_applicable( T, (int)D, 2 * (int)cfg_.hidden_dim ) )
                {
                    w4a8 = true;      // (below: never the bf16 fused form)
                    if ( mila_cdna4_gemm_geglu_w4a8_applicable( T, (int)D, (int)cfg_.hidden_dim ) )
                    {
                        const auto* W8 = static_cast<const uint8_t*>( L.fc_gate_up->getWeight().rawData() );
                        const float* sc = L.fc_gate_up->getWeightScale()->data();
                        if ( x8_in && ts_in )
                        {
                            Compute::rocmCheck( mila_cdna4_gemm_geglu_fp8_w8a8( act.data(), x8_in, W8, ts_in, sc, T, (int)D, (int)cfg_.hidden_dim, st ) );
                            return;
                        }
                        uint8_t* x8; float* ts;
                        op.activationScratch( T, (int)D, x8, ts );
                        Compute::rocmCheck( mila_cdna4_quantize_fp8_per_token( x8, ts, ffn_in.data(), T, (int)D, st ) );
                        Compute::rocmCheck( mila_cdna4_gemm_geglu_fp8_w8a8( act.data(), x8, W8, ts, sc, T, (int)D, (int)cfg_.hidden_dim, st ) );
                        return;
                    }
                }
            }
            if constexpr ( kFmt == 2 )
            {
                if ( w4a8 && mila_cdna4_gemm_geglu_w4a8_applicable( T, (int)D, (int)cfg_.hidden_dim ) && L.fc_gate_up->getOperation().weightFp8Scale() &&
                     L.fc_gate_up->getOperation().residentE4m3() )
                {
                    // resident e4m3 weights (staged once at load): only the activations are quantized per forward
                    auto& op = L.fc_gate_up->getOperation();
                    if ( x8_in && ts_in )
                    {
                        Compute::rocmCheck( mila_cdna4_gemm_geglu_fp8_scaled( act.data(), x8_in, op.residentE4m3(), ts_in, op.weightFp8Scale(), T, (int)D, (int)cfg_.hidden_dim, st ) );
                        return;
                    }
                    uint8_t* x8; float* ts;
                    op.activationScratch( T, (int)D, x8, ts );
                    Compute::rocmCheck( mila_cdna4_quantize_fp8_per_token( x8, ts, ffn_in.data(), T, (int)D, st ) );
                    Compute::rocmCheck( mila_cdna4_gemm_geglu_fp8_scaled( act.data(), x8, op.residentE4m3(), ts, op.weightFp8Scale(), T, (int)D, (int)cfg_.hidden_dim, st ) );
                    return;
                }
                if ( w4a8 && mila_cdna4_gemm_geglu_w4a8_applicable( T, (int)D, (int)cfg_.hidden_dim ) && L.fc_gate_up->getOperation().weightFp8Scale() )
                {
                    const int F = (int)cfg_.hidden_dim;
                    const size_t need = mila_cdna4_gemm_w4a8_scratch_bytes( T, (int)D, 2 * F );
                    void* scratch = ctx_->getScratch( need );
                    Compute::rocmCheck( mila_cdna4_gemm_geglu_bf16_w4a8( act.data(), ffn_in.data(), static_cast<const uint8_t*>( L.fc_gate_up->getWeight().rawData() ),
                                                                         L.fc_gate_up->getWeightScale()->data(), L.fc_gate_up->getOperation().weightFp8Scale(), T, (int)D, F,
                                                                         Quant::Weight::groupSizeOf<TWeightQuant>(), scratch, need, st ) );
                    return;
                }
            }
            if ( !w4a8 && mila_cdna4_gemm_geglu_preferred( T, (int)D, (int)cfg_.hidden_dim ) )      // (this caller holds the split-K workspace)
            {
                // Linear + GeGLU in one kernel: the [T, 2F] gate|up intermediate never reaches memory (bit-identical to the pair)
                const int F = (int)cfg_.hidden_dim;
                const void* W = L.fc_gate_up->getWeight().rawData();
                if constexpr ( kFmt == 0 )
                    Compute::rocmCheck( mila_cdna4_gemm_geglu_bf16( act.data(), ffn_in.data(), static_cast<const uint16_t*>( W ), T, (int)D, F, st ) );
                else if ( kFmt == 1 && L.fc_gate_up->getOperation().residentBf16() )
                    Compute::rocmCheck( mila_cdna4_gemm_geglu_bf16( act.data(), ffn_in.data(), L.fc_gate_up->getOperation().residentBf16(), T, (int)D, F, st ) );
                else
                {
                    const size_t need = (size_t)2 * F * D * 2;
                    void* scratch = ctx_->getScratch( need );
                    if constexpr ( kFmt == 1 )
                        Compute::rocmCheck( mila_cdna4_gemm_geglu_bf16_w8a16_staged( act.data(), ffn_in.data(), static_cast<const uint8_t*>( W ), L.fc_gate_up->getWeightScale()->data(),
                                                                                     T, (int)D, F, scratch, need, st ) );
                    else
                        Compute::rocmCheck( mila_cdna4_gemm_geglu_bf16_w4a16_staged( act.data(), ffn_in.data(), static_cast<const uint8_t*>( W ), L.fc_gate_up->getWeightScale()->data(),
                                                                                     T, (int)D, F, Quant::Weight::groupSizeOf<TWeightQuant>(), scratch, need, st ) );
                }
            }
            else
            {
                // the unfused pair; a caller running two calls at once (halfBlock) hands each its own [T, 2F] rows instead of the component's output
                if ( private_gate_up ) L.fc_gate_up->getOperation().forward( ffn_in, *private_gate_up );
                auto& gate_up = private_gate_up ? *private_gate_up : L.fc_gate_up->forward( ffn_in );
                Compute::rocmCheck( mila_cdna4_geglu_bf16( act.data(), gate_up.data(), T, (int)cfg_.hidden_dim, st ) );
            }
        }

        /// GemmaBlock::forward with the glue fused (bit-identical to GemmaBlock::prefill): the packed qkv rows go straight through
        /// q/k/v norm + RoPE into q and the KV cache (no split3 / kv_write), and each sandwich tail (RmsNorm + Residual
        /// (+ layer scalar) + the next RmsNorm) is one launch.  `have_normed`: the previous block's tail already wrote
        /// input_norm(input) into pf_norm_; `nextL`: the block whose input_norm the second tail applies.
        void blockPrefillFused( Layer& L, TensorType& input, bool have_normed, TensorType& output, Layer* nextL, int T, int position_offset )
        {
            const bool g = L.global;
            const dim_t NH = cfg_.num_heads, NKV = cfg_.numKvHeads( g ), HD = cfg_.headDim( g ), D = cfg_.embedding_dim;
            mila_stream_t st = ctx_->getStream();
            auto x3 = input.view( shape_t{ 1, T, D } );
            auto normed_view = pf_norm_->view( shape_t{ 1, T, D } );
            TensorType* normed = &normed_view;
            if ( !have_normed ) { normed = &L.input_norm->forward( x3 ); pf_q8_normed_ = false; }
            // W4A8 policy: the previous block's second tail wrote these rows quantized as well -- the Linear's own quantization launch is skipped (same bits)
            bool q8_in = false;
            if constexpr ( kFmt != 0 ) q8_in = have_normed && pf_q8_normed_ && L.qkv_proj->getOperation().acceptsFp8Activations( T );
            auto& qkv = q8_in ? L.qkv_proj->forwardFp8Activations( static_cast<const uint8_t*>( pf_q8_[ 1 ]->rawData() ), pf_ts_[ 1 ]->data(), normed->shape() )
                              : L.qkv_proj->forward( *normed );
            auto q = q_->view( shape_t{ 1, T, NH * HD } );
            const uint16_t* qp = static_cast<const uint16_t*>( qkv.rawData() );
            const uint16_t* kp = qp + (size_t)( NH * HD );
            const uint16_t* vp = g ? kp : kp + (size_t)( NKV * HD );
            Compute::rocmCheck( mila_cdna4_fused_qkv_post_prefill( q.data(), L.keyCache(), L.valueCache(), qp, kp, vp, (int64_t)cfg_.packedQkvWidth( g ),
                                                                   L.q_norm->getWeight()->data(), L.k_norm->getWeight()->data(), L.v_norm->getWeight()->data(),
                                                                   L.rope->cosCache(), L.rope->sinCache(), T, (int)NH, (int)NKV, (int)HD, position_offset,
                                                                   (int)L.cacheCapacity(), cfg_.rms_norm_eps, st ) );
            auto attn = attn_out_->view( shape_t{ 1, T, NH * HD } );
            L.prefillFromCache( q, attn, T, position_offset );
            auto& o = L.o_proj->forward( attn );
            auto res1 = res1_->view( shape_t{ 1, T, D } );
            auto ffn_in = pf_norm2_->view( shape_t{ 1, T, D } );
            // W4A8 / W8A8: a tail whose normalised rows feed a Linear on the fp8 x fp8 path writes them quantized per token as well (fused_tail_norm_quant)
            bool q8_ffn = false, q8_next = false;
            if constexpr ( kFmt != 0 )
            {
                auto& gu = L.fc_gate_up->getOperation();
                q8_ffn = gu.acceptsFp8Activations( T ) && mila_cdna4_gemm_geglu_w4a8_applicable( T, (int)D, (int)cfg_.hidden_dim ) != 0;
                q8_next = nextL && nextL->qkv_proj->getOperation().acceptsFp8Activations( T );
            }
            if ( q8_ffn )
                Compute::rocmCheck( mila_cdna4_fused_tail_norm_quant_bf16( res1.data(), ffn_in.data(), static_cast<uint8_t*>( pf_q8_[ 0 ]->rawData() ), pf_ts_[ 0 ]->data(), o.data(), x3.data(),
                                                                           L.post_attn_norm->getWeight()->data(), L.pre_ffn_norm->getWeight()->data(), T, (int)D, 1.0f, cfg_.rms_norm_eps, st ) );
            else
                Compute::rocmCheck( mila_cdna4_fused_tail_norm_bf16( res1.data(), ffn_in.data(), o.data(), x3.data(), L.post_attn_norm->getWeight()->data(),
                                                                     L.pre_ffn_norm->getWeight()->data(), T, (int)D, 1.0f, cfg_.rms_norm_eps, st ) );
            auto act = geglu_->view( shape_t{ 1, T, cfg_.hidden_dim } );
            if ( q8_ffn ) gateUpGeglu( L, ffn_in, act, T, nullptr, static_cast<const uint8_t*>( pf_q8_[ 0 ]->rawData() ), pf_ts_[ 0 ]->data() );
            else gateUpGeglu( L, ffn_in, act, T );
            auto& ffn = L.fc_down->forward( act );
            if ( q8_next )
                Compute::rocmCheck( mila_cdna4_fused_tail_norm_quant_bf16( output.data(), pf_norm_->data(), static_cast<uint8_t*>( pf_q8_[ 1 ]->rawData() ), pf_ts_[ 1 ]->data(), ffn.data(), res1.data(),
                                                                           L.post_ffn_norm->getWeight()->data(), nextL->input_norm->getWeight()->data(), T, (int)D, L.layer_scalar,
                                                                           cfg_.rms_norm_eps, st ) );
            else
                Compute::rocmCheck( mila_cdna4_fused_tail_norm_bf16( output.data(), nextL ? pf_norm_->data() : nullptr, ffn.data(), res1.data(), L.post_ffn_norm->getWeight()->data(),
                                                                     nextL ? nextL->input_norm->getWeight()->data() : nullptr, T, (int)D, L.layer_scalar, cfg_.rms_norm_eps, st ) );
            pf_q8_normed_ = q8_next;
        }

        // ---- two halves of a chunk on two streams -------------------------------------------------------------------------------------------
        // A GEMM launch leaves CUs idle in its last round of tiles (fc_gate_up: 960 tiles on 256 CUs) and at every kernel boundary.  Rows are independent
        // but for attention, so the chunk's two halves run as two kernel sequences on two streams -- each on its own rows of every buffer -- and the second
        // half's attention of a layer waits (an event) for the first half's K / V rows of that layer.  The idle CUs of one stream's tails take the other
        // stream's workgroups.  Same kernels on the same rows: bit-identical to the one-stream form.
        bool overlapApplicable( dim_t T ) const
        {
            if constexpr ( kFmt == 2 ) return false;       // the W4A8 path quantizes activations into per-op scratch sized for one call at a time
            if constexpr ( kFmt == 1 ) { for ( auto& L : layers_ ) if ( L.fc_down->getOperation().fp8ActivationPrefill() || !L.fc_down->getOperation().residentBf16() ) return false; }
            if ( !( T >= 1024 && T % 512 == 0 ) ) return false;
            // ... and so does a GEMM that splits K through the context's workspace (short tile lists: the same idle CUs this form is after); the split also depends on
            // the row count, so the halves would not carry the whole chunk's bits
            for ( auto& L : layers_ )
            {
                const Compute::LinearOpConfig* cfgs[ 4 ] = { &L.qkv_proj->getOperation().config(), &L.o_proj->getOperation().config(), &L.fc_gate_up->getOperation().config(),
                                                             &L.fc_down->getOperation().config() };
                for ( const auto* c : cfgs )
                    for ( dim_t rows : { T, T / 2 } )
                        if ( mila_cdna4_gemm_workspace_bytes( static_cast<int>( rows ), static_cast<int>( c->in_features ), static_cast<int>( c->out_features ) ) != 0 ) return false;
            }
            return true;
        }
        void halfBlock( Layer& L, int h, bool first_layer, Layer* nextL, int H, int position_offset, int flip, hipEvent_t kv_ready, bool wait_kv )
        {
            const bool g = L.global;
            const dim_t NH = cfg_.num_heads, NKV = cfg_.numKvHeads( g ), HD = cfg_.headDim( g ), D = cfg_.embedding_dim, F = cfg_.hidden_dim;
            const size_t r0 = static_cast<size_t>( h ) * static_cast<size_t>( H );
            mila_stream_t st = ctx_->getStream();
            // a half's region of a buffer starts at its first row in the buffer's OWN (widest) row pitch: the first half may run a layer of the other kind
            // (wider q / qkv rows) ahead of the second half, and the regions of the two must not meet whatever the widths
            auto rows = [&]( TensorType& t, dim_t width ) { return t.slice( r0 * static_cast<size_t>( t.shape().back() ), shape_t{ 1, H, width } ); };
            auto x3 = rows( *pf_x_[ flip ], D );
            auto out = rows( *pf_x_[ 1 - flip ], D );
            auto normed = rows( *pf_norm_, D );
            (void)first_layer;
            auto qkv = rows( *ov_qkv_, cfg_.packedQkvWidth( g ) );
            L.qkv_proj->getOperation().forward( normed, qkv );
            auto q = rows( *q_, NH * HD );
            const uint16_t* qp = qkv.data();
            const uint16_t* kp = qp + (size_t)( NH * HD );
            const uint16_t* vp = g ? kp : kp + (size_t)( NKV * HD );
            Compute::rocmCheck( mila_cdna4_fused_qkv_post_prefill( q.data(), L.keyCache(), L.valueCache(), qp, kp, vp, (int64_t)cfg_.packedQkvWidth( g ),
                                                                   L.q_norm->getWeight()->data(), L.k_norm->getWeight()->data(), L.v_norm->getWeight()->data(),
                                                                   L.rope->cosCache(), L.rope->sinCache(), H, (int)NH, (int)NKV, (int)HD, position_offset + h * H,
                                                                   (int)L.cacheCapacity(), cfg_.rms_norm_eps, st ) );
            if ( h == 0 ) hipCheck( hipEventRecord( kv_ready, reinterpret_cast<hipStream_t>( st ) ), "hipEventRecord" );
            else if ( wait_kv ) hipCheck( hipStreamWaitEvent( reinterpret_cast<hipStream_t>( st ), kv_ready, 0 ), "hipStreamWaitEvent" );
            auto attn = rows( *attn_out_, NH * HD );
            L.prefillFromCache( q, attn, H, position_offset + h * H );
            auto o = rows( *ov_o_, D );
            L.o_proj->getOperation().forward( attn, o );
            auto res1 = rows( *res1_, D );
            auto ffn_in = rows( *pf_norm2_, D );
            Compute::rocmCheck( mila_cdna4_fused_tail_norm_bf16( res1.data(), ffn_in.data(), o.data(), x3.data(), L.post_attn_norm->getWeight()->data(),
                                                                 L.pre_ffn_norm->getWeight()->data(), H, (int)D, 1.0f, cfg_.rms_norm_eps, st ) );
            auto act = rows( *geglu_, F );
            auto gate_up = rows( *ov_gate_up_, 2 * F );        // only written when the fused Linear + GeGLU does not serve this [H, D, F]
            gateUpGeglu( L, ffn_in, act, H, &gate_up );
            auto ffn = rows( *ov_o_, D );                      // o is dead after the first tail
            L.fc_down->getOperation().forward( act, ffn );
            Compute::rocmCheck( mila_cdna4_fused_tail_norm_bf16( out.data(), nextL ? normed.data() : nullptr, ffn.data(), res1.data(), L.post_ffn_norm->getWeight()->data(),
                                                                 nextL ? nextL->input_norm->getWeight()->data() : nullptr, H, (int)D, L.layer_scalar, cfg_.rms_norm_eps, st ) );
        }
        LogitsTensor& prefillOverlapped( const TokenTensor& tokens, dim_t T, dim_t position_offset )
        {
            const dim_t D = cfg_.embedding_dim;
            const int H = static_cast<int>( T / 2 );
            const auto dev = ctx_->getDeviceId();
            if ( !ov_qkv_ )
            {
                ov_qkv_ = std::make_unique<TensorType>( dev, shape_t{ 1, max_prefill_, std::max( cfg_.packedQkvWidth( false ), cfg_.packedQkvWidth( true ) ) } );
                ov_o_ = std::make_unique<TensorType>( dev, shape_t{ 1, max_prefill_, D } );
                ov_gate_up_ = std::make_unique<TensorType>( dev, shape_t{ 1, max_prefill_, 2 * cfg_.hidden_dim } );
                hipCheck( hipStreamCreateWithFlags( &ov_stream_, hipStreamNonBlocking ), "hipStreamCreate" );
                for ( auto& e : ov_ev_ ) hipCheck( hipEventCreateWithFlags( &e, hipEventDisableTiming ), "hipEventCreate" );
            }
            hipStream_t main = reinterpret_cast<hipStream_t>( ctx_->getStream() );
            embed( tokens.data(), static_cast<int>( T ), *pf_x_[ 0 ] );
            {
                // the first block's input norm over the whole chunk, before the streams part (the op's rstd rows are per call)
                auto x_all = pf_x_[ 0 ]->view( shape_t{ 1, T, D } );
                auto n_all = pf_norm_->view( shape_t{ 1, T, D } );
                layers_[ 0 ].input_norm->getOperation().forward( x_all, n_all );
            }
            hipCheck( hipEventRecord( ov_ev_[ 0 ], main ), "hipEventRecord" );
            hipCheck( hipStreamWaitEvent( ov_stream_, ov_ev_[ 0 ], 0 ), "hipStreamWaitEvent" );
            int flip = 0;
            for ( size_t i = 0; i < layers_.size(); ++i )
            {
                Layer* nextL = i + 1 < layers_.size() ? &layers_[ i + 1 ] : nullptr;
                hipEvent_t kv = ov_ev_[ 1 + ( i % 6 ) ];
                halfBlock( layers_[ i ], 0, i == 0, nextL, H, static_cast<int>( position_offset ), flip, kv, false );
                {
                    struct Scope { Compute::RocmExecutionContext* c; mila_stream_t old; ~Scope() { c->swapStream( old ); } } scope{ ctx_, ctx_->swapStream( reinterpret_cast<mila_stream_t>( ov_stream_ ) ) };
                    halfBlock( layers_[ i ], 1, i == 0, nextL, H, static_cast<int>( position_offset ), flip, kv, true );
                }
                flip = 1 - flip;
            }
            hipCheck( hipEventRecord( ov_ev_[ 7 ], ov_stream_ ), "hipEventRecord" );
            hipCheck( hipStreamWaitEvent( main, ov_ev_[ 7 ], 0 ), "hipStreamWaitEvent" );
            auto last = pf_x_[ flip ]->slice( static_cast<size_t>( ( T - 1 ) * D ), shape_t{ 1, 1, D } );
            auto& normed = final_norm_->forward( last );
            head( normed.data() );
            return *logits_;
        }

    public:
        /// two half-chunks on two streams (see halfBlock): off by default until measured per deployment; same bits either way
        void setPrefillOverlap( bool on ) { prefill_overlap_ = on; }
        /// the fused prefill glue serves 1024 < D <= 8192 (workgroup-per-row canonical RMS reduction)
        bool fusedPrefillApplicable() const { return cfg_.embedding_dim > 1024 && cfg_.embedding_dim <= 8192 && cfg_.embedding_dim % 8 == 0; }
        /// on (default): prefill runs the fused glue when the configuration fits; off: one launch per reference op.  Same bits.
        void setFusedPrefill( bool on ) { fused_prefill_ = on; }
        /// fp4 policy: W4A8 prefill (fp4 -> e4m3 weights, per-token e4m3 activations, fp8 MFMA; the reference's default) on every
        /// layer Linear, or the exact-weight fallback (dequantize -> bf16 MFMA).
        /// fp8 policy: W8A8 prefill (the policy's e4m3 weights + per-channel scales on the fp8 matrix cores, no resident bf16 copy; default off -- the reference's
        /// arithmetic for this policy is W8A16) on every layer Linear.  No effect on unquantized weights.
        void setFp8ActivationPrefill( bool on )
        {
            if constexpr ( kFmt != 0 )
                for ( auto& L : layers_ )
                {
                    L.qkv_proj->getOperation().setFp8ActivationPrefill( on ); L.o_proj->getOperation().setFp8ActivationPrefill( on );
                    L.fc_gate_up->getOperation().setFp8ActivationPrefill( on ); L.fc_down->getOperation().setFp8ActivationPrefill( on );
                }
        }

    private:
        // ---- fused step ---------------------------------------------------------------------------------
        mila_fused_matvec_args baseArgs( LinearType& lin, uint16_t* y, const uint16_t* x ) const
        {
            mila_fused_matvec_args a{};
            a.y = y; a.x = x; a.W = lin.getWeight().rawData();
            a.scales = nullptr;
            if constexpr ( TWeightQuant::kIsQuantized ) a.scales = lin.getWeightScale()->data();
            a.post_scale = 1.0f; a.eps = cfg_.rms_norm_eps; a.fmt = kFmt; a.K = (int)lin.getConfig().getInputFeatures();
            a.N = (int)lin.getConfig().getOutputFeatures(); a.group = Quant::Weight::groupSizeOf<TWeightQuant>(); a.geglu = 0;
            return a;
        }
        void plainMatvec( LinearType& lin, uint16_t* y, const uint16_t* x )
        {
            auto in = TensorType();   // route through the op so the launch is the same C-ABI call as Linear::forward
            (void)in;
            const int K = (int)lin.getConfig().getInputFeatures(), N = (int)lin.getConfig().getOutputFeatures();
            mila_stream_t st = ctx_->getStream();
            if constexpr ( kFmt == 0 ) Compute::rocmCheck( mila_cdna4_matvec_bf16( y, x, static_cast<const uint16_t*>( lin.getWeight().rawData() ), nullptr, K, N, st ) );
            else if constexpr ( kFmt == 1 ) Compute::rocmCheck( mila_cdna4_matvec_bf16_qfp8( y, x, static_cast<const uint8_t*>( lin.getWeight().rawData() ), lin.getWeightScale()->data(), nullptr, K, N, st ) );
            else Compute::rocmCheck( mila_cdna4_matvec_bf16_qfp4( y, x, static_cast<const uint8_t*>( lin.getWeight().rawData() ), lin.getWeightScale()->data(), nullptr, K, N,
                                                                  Quant::Weight::groupSizeOf<TWeightQuant>(), st ) );
        }
        void fusedGateUp( Layer& L )
        {
            auto a = baseArgs( *L.fc_gate_up, f_act_->data(), f_o_->data() );
            a.N = (int)cfg_.hidden_dim; a.geglu = 1;
            a.norm_w = L.pre_ffn_norm->getWeight()->data(); a.post_w = L.post_attn_norm->getWeight()->data();
            a.res = cur_hidden_; a.res_out = res1_->data();
            Compute::rocmCheck( mila_cdna4_fused_norm_matvec( &a, ctx_->getStream() ) );
        }

        /// x_l (hidden) -> x_{l+1}.  The sandwich tail of layer l-1 (post_ffn_norm, residual, layer scalar)
        /// is the prologue of layer l's qkv kernel; the tail of the last layer is the prologue of the head.
        /// sampler_partials != nullptr: the head also runs the greedy sampler's first stage into sample_scratch_ and reports the number of partials
        void enqueueFusedStep( const int32_t* token_dev, int position, const int32_t* pos_dev, int* sampler_partials = nullptr )
        {
            mila_stream_t st = ctx_->getStream();
            embed( token_dev, 1, *hidden_[ 0 ] );
            cur_hidden_ = hidden_[ 0 ]->data();
            int next = 1;
            const Layer* prev = nullptr;
            for ( auto& L : layers_ )
            {
                const bool g = L.global;
                const int NH = (int)cfg_.num_heads, NKV = (int)cfg_.numKvHeads( g ), HD = (int)cfg_.headDim( g );
                // 1. [tail of previous layer] + input_norm + qkv projection
                auto a = baseArgs( *L.qkv_proj, f_qkv_->data(), prev ? f_down_->data() : cur_hidden_ );
                a.norm_w = L.input_norm->getWeight()->data();
                if ( prev )
                {
                    a.post_w = prev->post_ffn_norm->getWeight()->data(); a.res = res1_->data(); a.res_out = hidden_[ next ]->data(); a.post_scale = prev->layer_scalar;
                }
                Compute::rocmCheck( mila_cdna4_fused_norm_matvec( &a, st ) );
                if ( prev ) { cur_hidden_ = hidden_[ next ]->data(); next = ( next == 1 ) ? 2 : 1; }
                // 2+3. q/k/v norms + RoPE + KV append + flash-decode in one launch (+ combine).
                //        qkv row = [q | k | v] (global: [q | k], V from the raw k projection)
                const uint16_t* qp = f_qkv_->data();
                const uint16_t* kp = qp + (size_t)NH * HD;
                const uint16_t* vp = g ? kp : kp + (size_t)NKV * HD;
                const size_t need = attnScratchBytes();
                void* scratch = attn_partials_->data();
                Compute::rocmCheck( mila_cdna4_fused_attn_decode_bf16( attn_out_->data(), L.keyCache(), L.valueCache(), qp, kp, vp, L.q_norm->getWeight()->data(),
                                                                       L.k_norm->getWeight()->data(), L.v_norm->getWeight()->data(), L.rope->cosCache(), L.rope->sinCache(),
                                                                       scratch, need, NH, NKV, HD, (int)L.cacheCapacity(), position, pos_dev,
                                                                       (int)cfg_.windowFor( g ), L.attentionScale(), cfg_.rms_norm_eps, st ) );
                // 4. o_proj
                plainMatvec( *L.o_proj, f_o_->data(), attn_out_->data() );
                // 5. post_attn_norm + residual + pre_ffn_norm + gate_up + GeGLU
                fusedGateUp( L );
                // 6. fc_down
                plainMatvec( *L.fc_down, f_down_->data(), f_act_->data() );
                prev = &L;
            }
            // tail of the last layer + final norm + lm_head (fp32 logits) in one launch
            {
                mila_fused_matvec_args a{};
                a.y = reinterpret_cast<uint16_t*>( logits_->data() ); a.x = f_down_->data(); a.W = lm_head_->getWeight().rawData();
                a.scales = nullptr;
                if constexpr ( kTableFmt != 0 ) a.scales = lm_head_->getWeightScale()->data();
                a.norm_w = final_norm_->getWeight()->data(); a.post_w = prev->post_ffn_norm->getWeight()->data();
                a.res = res1_->data(); a.res_out = hidden_[ next ]->data(); a.post_scale = prev->layer_scalar; a.eps = cfg_.rms_norm_eps;
                a.fmt = kTableFmt; a.K = (int)cfg_.embedding_dim; a.N = (int)cfg_.vocab_size; a.group = 0; a.geglu = 0; a.f32_out = 1;
                if ( sampler_partials ) { a.argmax_scratch = sample_scratch_->data(); a.argmax_scratch_bytes = sample_scratch_->sizeInBytes(); a.argmax_blocks = sampler_partials; }
                Compute::rocmCheck( mila_cdna4_fused_norm_matvec( &a, st ) );
            }
        }

    public:
        // ------------------------------------------------------------------------------------
        // weight ingestion.  Tensor vocabulary = the reference's flat names (the root's own name dropped,
        // Core/LanguageModel.ixx:137-146): `tf_layer_<i>.<child>.weight` (+ `.weight_scale` for a quantized Linear, Linear.ixx:370-400),
        // `tf_layer_<i>.layer_scalar` ([1] F32, Gemma.Block.ixx:546-560), `temb.wte` (+ `temb.wte_scale`, TokenEmbedding.ixx:351-384),
        // `rmsn_final.weight`; a tied model has NO `lm_head.weight` (Gemma.ixx:540-555) -- the head adopts the embedding table.
        // Readers accept the names with or without the `<model name>.` prefix (CompositeComponent::findComponent strips it).
        // ------------------------------------------------------------------------------------
        static const char* weightQuantizationName() { return kFmt == 0 ? "none" : kFmt == 1 ? "per_channel_fp8_e4m3" : "per_group_fp4_128"; }   // LanguageModelConfig.ixx:104-114

        Serialization::PretrainedMetadata pretrainedMetadata() const
        {
            Serialization::PretrainedMetadata m;
            m.architecture = "gemma4"; m.model_name = name_;
            m.vocab_size = (uint32_t)cfg_.vocab_size; m.max_seq_length = (uint32_t)max_seq_; m.embedding_dim = (uint32_t)cfg_.embedding_dim; m.num_layers = (uint32_t)cfg_.num_layers;
            m.num_heads = (uint32_t)cfg_.num_heads; m.num_kv_heads = (uint32_t)cfg_.num_kv_heads; m.head_dim = (uint32_t)cfg_.head_dim; m.hidden_dim = (uint32_t)cfg_.hidden_dim;
            m.use_bias = false; m.tie_word_embeddings = true; m.activation = "gelu"; m.norm_type = "rmsnorm"; m.attention_type = "gqa"; m.positional_encoding = "rope";
            m.rope_theta = cfg_.rope_theta_local; m.norm_epsilon = cfg_.rms_norm_eps;
            m.global_head_dim = (uint32_t)cfg_.global_head_dim; m.num_global_kv_heads = (uint32_t)cfg_.num_global_kv_heads; m.key_equals_value = true;
            m.window = (uint32_t)cfg_.window; m.sliding_window_pattern = (uint32_t)cfg_.sliding_window_pattern; m.global_rotary_dim = (uint32_t)cfg_.global_rotary_dim;
            m.rope_theta_local = cfg_.rope_theta_local; m.rope_theta_global = cfg_.rope_theta_global; m.final_logit_softcapping = cfg_.final_logit_softcapping;
            return m;
        }

    private:
        struct SaveItem { std::string name, dtype; std::vector<int64_t> shape; const void* dev; size_t bytes; float host_scalar; };
        /// the model's tensors in the reference's declaration order (block children in createGraph order, then the block's own scalar)
        std::vector<SaveItem> saveItems()
        {
            std::vector<SaveItem> items;
            auto shapeOf = []( const auto& t ) { std::vector<int64_t> v; for ( auto d : t.shape() ) v.push_back( (int64_t)d ); return v; };
            auto storageName = []( auto& lin )
            {
                using L = std::remove_reference_t<decltype( lin )>;
                if constexpr ( !L::kIsQuantized ) return "BF16"; else return L::kWeightDtype == TensorDataType::FP8_E4M3 ? "F8_E4M3" : "U8";
            };
            auto addLinear = [&]( auto& lin, const std::string& flat, const char* wname, const char* sname )
            {
                auto& w = lin.getWeight();
                items.push_back( { flat + "." + wname, storageName( lin ), shapeOf( w ), w.rawData(), w.sizeInBytes(), 0.0f } );
                if constexpr ( std::remove_reference_t<decltype( lin )>::kIsQuantized )
                    items.push_back( { flat + "." + sname, "F32", shapeOf( *lin.getWeightScale() ), lin.getWeightScale()->rawData(), lin.getWeightScale()->sizeInBytes(), 0.0f } );
            };
            auto addNorm = [&]( RmsNormType& n ) { items.push_back( { flatName( n.getName() ) + ".weight", "BF16", shapeOf( *n.getWeight() ), n.getWeight()->rawData(), n.getWeight()->sizeInBytes(), 0.0f } ); };
            addLinear( *lm_head_, "temb", "wte", "wte_scale" );      // the tied table under the embedding's name; no lm_head.weight
            for ( size_t i = 0; i < layers_.size(); ++i )
            {
                auto& L = layers_[ i ];
                addNorm( *L.input_norm ); addNorm( *L.q_norm ); addNorm( *L.k_norm ); addNorm( *L.v_norm ); addNorm( *L.post_attn_norm ); addNorm( *L.pre_ffn_norm ); addNorm( *L.post_ffn_norm );
                addLinear( *L.qkv_proj, flatName( L.qkv_proj->getName() ), "weight", "weight_scale" ); addLinear( *L.o_proj, flatName( L.o_proj->getName() ), "weight", "weight_scale" );
                addLinear( *L.fc_gate_up, flatName( L.fc_gate_up->getName() ), "weight", "weight_scale" ); addLinear( *L.fc_down, flatName( L.fc_down->getName() ), "weight", "weight_scale" );
                items.push_back( { "tf_layer_" + std::to_string( i ) + ".layer_scalar", "F32", { 1 }, nullptr, 4, L.layer_scalar } );
            }
            addNorm( *final_norm_ );
            return items;
        }
        std::string flatName( const std::string& component_name ) const
        {
            return component_name.compare( 0, name_.size() + 1, name_ + "." ) == 0 ? component_name.substr( name_.size() + 1 ) : component_name;
        }
        template<typename Writer> void writeItems( Writer& w, std::vector<SaveItem>& items )
        {
            w.beginData();
            std::vector<unsigned char> host;
            ctx_->synchronize();
            for ( auto& it : items )
            {
                if ( !it.dev ) { w.writeTensorData( it.name, &it.host_scalar, 4 ); continue; }
                host.resize( it.bytes );
                Compute::rocmCheck( mila_cdna4_memcpy_d2h( host.data(), it.dev, it.bytes, ctx_->getStream() ) );
                ctx_->synchronize();
                w.writeTensorData( it.name, host.data(), it.bytes );
            }
            w.close();
        }

    public:
        /// LanguageModel::savePretrained (Core/LanguageModel.ixx:116-148): every parameter in its STORAGE form (bf16, or e4m3 / packed e2m1 +
        /// fp32 scales) -- a quantized model writes a pre-quantized artifact that reloads without re-quantizing -- with the model description
        /// under __metadata__["mila_config"] and the policy under ["mila_quantization"]
        void saveSafeTensors( const std::string& path )
        {
            auto items = saveItems();
            Serialization::SafeTensorsWriter w( path );
            for ( auto& it : items ) w.declareTensor( it.name, it.dtype, it.shape );
            w.setMetadata( "format", "pt" );
            w.setMetadata( Serialization::kMilaQuantizationMetadataKey, weightQuantizationName() );
            w.setMetadata( Serialization::kMilaConfigMetadataKey, Serialization::toMetadataJSON( pretrainedMetadata() ) );
            writeItems( w, items );
        }
        /// the same tensors in the MILA .bin container (what the reference's converters emit and fromPretrained streams)
        void saveMilaBin( const std::string& path )
        {
            auto items = saveItems();
            Serialization::MilaBinWriter w( path );
            for ( auto& it : items ) w.declareTensor( it.name, it.dtype, it.shape );
            w.setMetadataJSON( Serialization::toMetadataJSON( pretrainedMetadata() ) );
            writeItems( w, items );
        }

        /// GemmaTransformer::loadParameters (Gemma.ixx:502-557): stream every blob of a MILA .bin or SafeTensors artifact in ascending file
        /// offset order into the component its flat name resolves to.  A Linear's `.weight` may be bf16 [N, K] (stored as is, or quantized
        /// on load under a quantized policy: Linear.ixx:529-558) or already in the policy's storage form with its `.weight_scale` sibling
        /// (:559-574); likewise `temb.wte` / `temb.wte_scale`.  The head is tied: it adopts the table, and an artifact that carries its own
        /// `lm_head.weight` is refused.  Unknown names, missing parameters, a geometry or policy that does not match this model: errors.
        void loadPretrained( const std::string& path )
        {
            destroyGraph();   // layer scalars are baked into the captured launches
            Compute::TraceRange tr( "gemma.loadPretrained" );
            Serialization::PretrainedModelReader r( path );
            const auto& md = r.getPretrainedMetadata();
            if ( !r.metadataJSON().empty() )
            {
                auto mismatch = [&]( const char* what, uint64_t file, uint64_t mine ) { if ( file != 0 && file != mine ) throw std::invalid_argument( "GemmaTransformer::loadPretrained: '" + path + "' declares " + what + " = " + std::to_string( file ) + ", this model has " + std::to_string( mine ) ); };
                mismatch( "vocab_size", md.vocab_size, (uint64_t)cfg_.vocab_size ); mismatch( "embedding_dim", md.embedding_dim, (uint64_t)cfg_.embedding_dim );
                mismatch( "num_layers", md.num_layers, (uint64_t)cfg_.num_layers ); mismatch( "hidden_dim", md.hidden_dim, (uint64_t)cfg_.hidden_dim );
                mismatch( "num_heads", md.num_heads, (uint64_t)cfg_.num_heads ); mismatch( "head_dim", md.head_dim, (uint64_t)cfg_.head_dim );
            }
            // a pre-quantized artifact loads only under the policy it was written with (GemmaModel.ixx:617-632)
            if ( !r.getWeightQuantization().empty() && r.getWeightQuantization() != weightQuantizationName() )
                throw std::runtime_error( "GemmaTransformer::loadPretrained: artifact '" + path + "' is pre-quantized as '" + r.getWeightQuantization() + "' but this model's policy is '" + weightQuantizationName() + "'" );
            using Entry = Serialization::SafeTensorsEntry;
            std::map<std::string, std::function<void( const Entry& )>> sinks;
            std::map<std::string, bool> required;
            auto bindLinear = [&]( auto& lin, const std::string& flat, const std::string& wname, const std::string& sname )
            {
                auto* lp = &lin;
                constexpr bool q = std::remove_reference_t<decltype( lin )>::kIsQuantized;
                sinks[ flat + "." + wname ] = [ lp ]( const Entry& e )
                {
                    const size_t NK = (size_t)lp->getConfig().getOutputFeatures() * lp->getConfig().getInputFeatures();
                    if ( e.dtype == "BF16" ) { if ( (size_t)e.elements() != NK ) throw std::invalid_argument( e.name + ": expected " + std::to_string( NK ) + " bf16 elements" ); }
                    else if ( !q || e.nbytes() != lp->getWeight().sizeInBytes() || ( e.dtype != "F8_E4M3" && e.dtype != "U8" ) )
                        throw std::invalid_argument( e.name + ": dtype " + e.dtype + " / " + std::to_string( e.nbytes() ) + " bytes does not fit this Linear's weight policy" );
                    lp->loadParameter( "weight", e.data, e.nbytes() );
                };
                required[ flat + "." + wname ] = false;
                if constexpr ( q )
                    sinks[ flat + "." + sname ] = [ lp ]( const Entry& e )
                    {
                        if ( e.dtype != "F32" ) throw std::invalid_argument( e.name + ": weight scales must be F32" );
                        lp->loadParameter( "weight_scale", e.data, e.nbytes() );
                    };
            };
            auto bindNorm = [&]( RmsNormType& n )
            {
                auto* np = &n;
                sinks[ flatName( n.getName() ) + ".weight" ] = [ np ]( const Entry& e )
                {
                    if ( e.dtype != "BF16" ) throw std::invalid_argument( e.name + ": norm weights must be BF16" );
                    np->loadParameter( "weight", e.data, e.nbytes() );
                };
                required[ flatName( n.getName() ) + ".weight" ] = false;
            };
            bindLinear( *lm_head_, "temb", "wte", "wte_scale" );
            for ( size_t i = 0; i < layers_.size(); ++i )
            {
                auto& L = layers_[ i ];
                bindNorm( *L.input_norm ); bindNorm( *L.q_norm ); bindNorm( *L.k_norm ); bindNorm( *L.v_norm ); bindNorm( *L.post_attn_norm ); bindNorm( *L.pre_ffn_norm ); bindNorm( *L.post_ffn_norm );
                for ( auto* lin : { L.qkv_proj.get(), L.o_proj.get(), L.fc_gate_up.get(), L.fc_down.get() } ) bindLinear( *lin, flatName( lin->getName() ), "weight", "weight_scale" );
                Layer* lp = &L;
                const std::string sn = "tf_layer_" + std::to_string( i ) + ".layer_scalar";
                sinks[ sn ] = [ lp ]( const Entry& e )
                {
                    if ( e.dtype != "F32" || e.elements() != 1 ) throw std::invalid_argument( e.name + ": layer_scalar must be one F32" );
                    std::memcpy( &lp->layer_scalar, e.data, 4 );
                };
                required[ sn ] = false;
            }
            bindNorm( *final_norm_ );
            std::vector<std::pair<std::string, std::string>> packed;     // storage-form tensor -> the scale sibling it needs
            const std::string prefix = name_ + ".";
            r.streamTensorBlobs( [&]( const std::string& full, const Entry& e )
            {
                const std::string flat = full.compare( 0, prefix.size(), prefix ) == 0 ? full.substr( prefix.size() ) : full;
                if ( flat == "lm_head.weight" || flat == "lm_head.weight_scale" )
                    throw std::invalid_argument( "GemmaTransformer::loadPretrained: '" + path + "' carries an untied '" + flat + "'; Gemma-4 ties the head to temb.wte (Gemma.ixx:540-555)" );
                auto it = sinks.find( flat );
                if ( it == sinks.end() ) throw std::invalid_argument( "GemmaTransformer::loadPretrained: '" + path + "' holds an unknown tensor '" + full + "'" );
                it->second( e );
                if ( required.count( flat ) ) required[ flat ] = true;
                const bool is_w = flat.size() > 7 && flat.compare( flat.size() - 7, 7, ".weight" ) == 0, is_t = flat == "temb.wte";
                if ( ( is_w || is_t ) && e.dtype != "BF16" ) packed.emplace_back( full, full + "_scale" );
            } );
            for ( auto& [ n, seen ] : required ) if ( !seen ) throw std::invalid_argument( "GemmaTransformer::loadPretrained: '" + path + "' lacks '" + n + "'" );
            for ( auto& [ n, sc ] : packed ) if ( !r.hasTensor( sc ) ) throw std::invalid_argument( "GemmaTransformer::loadPretrained: '" + n + "' is in storage form but '" + sc + "' is missing" );
            // op-owned derived state (fp4 tensor scale, resident prefill weights) once every sibling is in place, whatever the file order
            if constexpr ( TWeightQuant::kIsQuantized )
                for ( auto& L : layers_ )
                    for ( auto* lin : { L.qkv_proj.get(), L.o_proj.get(), L.fc_gate_up.get(), L.fc_down.get() } ) lin->getOperation().onQuantizedWeightsLoaded();
            ctx_->synchronize();
        }
        void loadSafeTensors( const std::string& path ) { loadPretrained( path ); }

        /// quantized policies: keep the prefill staging (fp8 -> bf16, fp4 -> e4m3) of every layer Linear resident (default on: +2 / +1
        /// bytes per weight of the 288 GB) or re-stage into scratch on every forward as the reference does; same bits
        void setResidentPrefillWeights( bool on )
        {
            for ( auto& L : layers_ )
                for ( auto* lin : { L.qkv_proj.get(), L.o_proj.get(), L.fc_gate_up.get(), L.fc_down.get() } ) lin->getOperation().setResidentPrefillWeights( on );
            ctx_->synchronize();
        }
        // ------------------------------------------------------------------------------------
        // footprint (Gemma.ixx:340-470): getMemoryStats() = what the model holds, getRequiredMemory() = what a model of this configuration, sequence length and
        // prefill chunk would hold once built and loaded -- computed from the configuration alone.  Two corrections as in the reference: the tied head reports the
        // shared table, subtracted once; every block reports one owner's RoPE tables, of which one pair per distinct geometry exists.
        // ------------------------------------------------------------------------------------
        MemoryStats getMemoryStats() const
        {
            MemoryStats st;
            std::map<const void*, std::pair<size_t, size_t>> rope;      // table -> (blocks sharing it, bytes one owner reports)
            for ( auto& L : layers_ )
            {
                st += L.getMemoryStats();
                auto& e = rope[ L.rope->getOperation().tableKey() ];
                e.first += 1; e.second = L.rope->getOperation().stateBytes();
            }
            for ( auto& [ key, e ] : rope ) st.device_state_bytes -= ( e.first - 1 ) * e.second;
            st += temb_->getMemoryStats();
            st += final_norm_->getMemoryStats();
            const MemoryStats head = lm_head_->getMemoryStats();
            st += head;
            st.device_parameter_bytes -= head.device_parameter_bytes;      // tied: the table is temb's
            st.device_state_bytes += ownStateBytes();
            return st;
        }
        MemoryStats getRequiredMemory() const
        {
            const dim_t D = cfg_.embedding_dim, P = max_prefill_;
            MemoryStats st;
            const BuildContext block_ctx( shape_t{ 1, P, D }, RuntimeMode::Inference );
            std::map<std::tuple<dim_t, uint32_t, dim_t>, std::pair<size_t, size_t>> rope;
            for ( auto& L : layers_ )
            {
                st += L.getRequiredMemory( block_ctx );
                uint32_t base_bits;
                const float base = L.rope->getConfig().getBase();
                std::memcpy( &base_bits, &base, 4 );
                auto& e = rope[ { L.rope->getConfig().getHeadDim(), base_bits, L.rope->getConfig().getRotaryDim() } ];
                e.first += 1; e.second = L.rope->getRequiredMemory( block_ctx ).device_state_bytes;
            }
            for ( auto& [ key, e ] : rope ) st.device_state_bytes -= ( e.first - 1 ) * e.second;
            st += temb_->getRequiredMemory( BuildContext( shape_t{ 1, P }, RuntimeMode::Inference ) );
            st += final_norm_->getRequiredMemory( BuildContext( shape_t{ 1, 1, D }, RuntimeMode::Inference ) );
            const MemoryStats head = lm_head_->getRequiredMemory( BuildContext( shape_t{ 1, 1, D }, RuntimeMode::Inference ) );
            st += head;
            st.device_parameter_bytes -= head.device_parameter_bytes;
            st.device_state_bytes += requiredOwnStateBytes();
            return st;
        }
    private:
        /// the transformer's own device buffers: the pooled block workspace, the prefill / fused-step scratch, logits, sampler and position words
        size_t ownStateBytes() const
        {
            size_t b = 0;
            for ( const auto& t : { q_, k_, v_, attn_out_, res1_, res2_, geglu_, blk_out_[ 0 ], blk_out_[ 1 ], ws_normed_, ws_o_normed_, ws_ffn_in_, ws_ffn_normed_, ws_q_normed_,
                                    ws_k_normed_, ws_v_normed_, ws_qkv_, ws_o_, ws_gate_up_, ws_down_ } ) b += tensorBytes( t );
            for ( const auto* t : { hidden_[ 0 ].get(), hidden_[ 1 ].get(), hidden_[ 2 ].get(), pf_x_[ 0 ].get(), pf_x_[ 1 ].get(), pf_norm_.get(), pf_norm2_.get(), pf_q8_[ 0 ].get(),
                                    pf_q8_[ 1 ].get(), f_qkv_.get(), f_q_.get(), f_o_.get(), f_down_.get(), f_act_.get(), ov_qkv_.get(), ov_o_.get(), ov_gate_up_.get() } ) b += tensorBytes( t );
            for ( const auto* t : { logits_.get(), sample_scratch_.get(), attn_partials_.get(), pf_ts_[ 0 ].get(), pf_ts_[ 1 ].get() } ) b += tensorBytes( t );
            b += tensorBytes( pos_dev_.get() );
            return b;
        }
        size_t requiredOwnStateBytes() const
        {
            const size_t D = static_cast<size_t>( cfg_.embedding_dim ), P = static_cast<size_t>( max_prefill_ ), F = static_cast<size_t>( cfg_.hidden_dim );
            const size_t maxq = static_cast<size_t>( std::max( cfg_.qWidth( false ), cfg_.qWidth( true ) ) ), maxkv = static_cast<size_t>( std::max( cfg_.kvWidth( false ), cfg_.kvWidth( true ) ) );
            const size_t maxpacked = static_cast<size_t>( std::max( cfg_.packedQkvWidth( false ), cfg_.packedQkvWidth( true ) ) );
            size_t e = 0;      // bf16 elements
            e += P * ( maxq + 2 * maxkv + maxq + D + D + F + 2 * D );                                       // q k v attn_out res1 res2 geglu blk_out x 2
            e += P * ( 4 * D + maxq + 2 * maxkv + maxpacked + D + 2 * F + D );                              // the pooled role outputs
            e += 3 * D + 2 * P * D + 2 * P * D;                                                              // hidden x 3, pf_x x 2, pf_norm, pf_norm2
            if ( kFmt != 0 ) e += 2 * P * ( ( D + 1 ) / 2 );                                                 // per-token e4m3 rows of the fp8 x fp8 prefill paths
            e += maxpacked + maxq + D + D + F;                                                               // f_qkv f_q f_o f_down f_act
            size_t b = e * 2;
            if ( ov_qkv_ ) b += P * ( maxpacked + D + 2 * F ) * 2;                                           // the two-stream prefill's buffers, once that path has run
            b += static_cast<size_t>( cfg_.vocab_size ) * 4 + ( ( mila_cdna4_sample_scratch_bytes() / 4 ) * 4 ) + ( ( attnScratchBytes() + 3 ) / 4 ) * 4;
            if ( kFmt != 0 ) b += 2 * P * 4;
            b += 4;                                                                                          // the device position word
            return b;
        }
    public:
        /// bytes of op-owned prefill staging the layer Linears hold right now (fp8 policy: bf16 copies, 0 while the W8A8 prefill is on; fp4 policy: e4m3 copies)
        double residentStagingBytes() const
        {
            double b = 0;
            for ( auto& L : layers_ )
                for ( auto* lin : { L.qkv_proj.get(), L.o_proj.get(), L.fc_gate_up.get(), L.fc_down.get() } ) b += static_cast<double>( lin->getOperation().residentBytes() );
            return b;
        }
    private:
        std::string name_{ "gemma" };      // the root's name (metadata.model_name in the reference); dropped from flat tensor names
        GemmaConfig cfg_;
        dim_t max_seq_, max_prefill_;
        std::unique_ptr<IExecutionContext> owned_ctx_;
        Compute::RocmExecutionContext* ctx_{ nullptr };
        LayerList layers_;
        std::shared_ptr<RmsNormType> final_norm_;
        std::shared_ptr<LmHeadLinearType> lm_head_;
        std::unique_ptr<TensorType> hidden_[ 3 ], pf_x_[ 2 ], f_qkv_, f_q_, f_o_, f_down_, f_act_;
        std::shared_ptr<TensorType> q_, k_, v_, attn_out_, res1_, res2_, geglu_, blk_out_[ 2 ];   // the blocks' shared workspace
        std::shared_ptr<TensorType> ws_normed_, ws_o_normed_, ws_ffn_in_, ws_ffn_normed_, ws_q_normed_, ws_k_normed_, ws_v_normed_, ws_qkv_, ws_o_, ws_gate_up_, ws_down_;   // ... one output per component role
        std::shared_ptr<TokenEmbeddingType> temb_;
        std::unique_ptr<LogitsTensor> logits_;
        std::unique_ptr<TokenTensor> pos_dev_;
        std::unique_ptr<LogitsTensor> sample_scratch_;
        bool sample_in_graph_{ false };
        bool fused_prefill_{ true };
        dim_t kv_fill_{ 0 };      ///< positions the caches hold, as far as eager calls tell (see rewindKvCache)
        bool prefill_overlap_{ false };
        std::unique_ptr<TensorType> ov_qkv_, ov_o_, ov_gate_up_;
        hipStream_t ov_stream_{ nullptr };
        hipEvent_t ov_ev_[ 8 ]{};
        std::unique_ptr<LogitsTensor> attn_partials_;
        std::unique_ptr<TensorType> pf_norm_, pf_norm2_;
        // W4A8 policy: the sandwich tails also write their normalised rows as per-token e4m3 + scales for the Linear that follows ([0]: pre_ffn_norm -> fc_gate_up,
        // [1]: the next block's input_norm -> its qkv_proj); model-owned like every prefill workspace
        std::unique_ptr<TensorType> pf_q8_[ 2 ];          // [1, P, D / 2] bf16 elements = P * D bytes of e4m3
        std::unique_ptr<LogitsTensor> pf_ts_[ 2 ];        // [P] fp32 per-token scales
        bool pf_q8_normed_{ false };       // pf_q8_[1] / pf_ts_[1] hold the rows of pf_norm_ (written by the previous block's second tail in this prefill)
        const uint16_t* cur_hidden_{ nullptr };
        hipGraph_t graph_{ nullptr };
        hipGraphExec_t graph_exec_{ nullptr };
        const int32_t* captured_token_{ nullptr };      // what the captured graph was built for: ensureGraph() re-captures on a mismatch
        dim_t captured_band_end_{ 0 };                  // ... and the live-length bucket (exclusive end) its attention launches were shaped for
        bool captured_sample_in_graph_{ false };
        unsigned long long* token_ring_{ nullptr };
        unsigned long long* token_seq_{ nullptr };
        const unsigned long long* captured_ring_{ nullptr };
        int token_ring_size_{ 0 };
    };
}
