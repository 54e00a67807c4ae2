// L6 caller of the hot path: GemmaModel -- load a pretrained artifact at a deployment configuration, then generate.  Mirrors
//   Models/GemmaModel.ixx:137-170 (fromPretrained -> dispatchWeightQuantization), :604-665 (fromPretrainedImpl: artifact / policy check, geometry from
//   the checkpoint metadata, context-length check, build, loadParameters), :439-568 (onGenerating: prompt check, stop set, transparent KV prefix reuse,
//   the decode-ahead pipeline, the four GenerateStatus outcomes), :750-770 (EOS / end-of-turn stop set, final logit softcap at the sampler);
//   Models/QuantizationDispatch.ixx:34-95; Core/LanguageModelConfig.ixx:88-230; Core/GenerateStatus.ixx; Components/Transformers/{GenerateParams,SamplingParams}.ixx.
// MI355X form of the decode-ahead pipeline: a greedy request runs the captured hipGraph whose last node is the device sampler (the next
// token never leaves the device) and publishes it into a host-visible ring (sequence number << 32 | token, one system-scope store); the host
// polls its slot while the next step is already running: no event, no copy, no stream wait on the decode path.  A stochastic request enqueues fused step + sampler per token (the uniform draw is a host scalar).
#pragma once

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <optional>
#include <random>
#include <span>
#include <unordered_set>
#include <variant>

#include "Gemma.h"

namespace Mila::Dnn
{
    enum class WeightQuantization { None, FP8, FP4 };
    enum class KvCacheCompression { None, FP8 };
    inline const char* weightQuantizationName( WeightQuantization wq )
    {
        switch ( wq ) { case WeightQuantization::FP8: return "per_channel_fp8_e4m3"; case WeightQuantization::FP4: return "per_group_fp4_128"; default: return "none"; }
    }

    enum class [[nodiscard]] GenerateStatus : int32_t { Success = 0, MaxNewTokensReached, ContextOverflow, ClientCancelled };
    inline std::string_view to_string( GenerateStatus s )
    {
        switch ( s )
        {
            case GenerateStatus::Success: return "stop";
            case GenerateStatus::MaxNewTokensReached: return "length";
            case GenerateStatus::ContextOverflow: return "context_limit";
            case GenerateStatus::ClientCancelled: return "cancelled";
        }
        return "unknown";
    }

    struct SamplingParams
    {
        float temperature = 1.0f;
        int top_k = 0;        ///< 0 disables top-k truncation; 1 == greedy
        float top_p = 1.0f;   ///< 1.0 disables nucleus truncation
    };
    struct GenerateParams
    {
        std::optional<int> max_new_tokens;       ///< nullopt => run to EOS / the context bound
        SamplingParams sampling{};
        std::vector<int32_t> stop_tokens{};      ///< empty => the model's default stop set
    };

    struct GemmaModelConfig
    {
        GemmaModelConfig() = default;
        explicit GemmaModelConfig( dim_t context_length ) { withContextLength( context_length ); }
        GemmaModelConfig& withContextLength( dim_t n )
        {
            if ( n <= 0 ) throw std::invalid_argument( "LanguageModelConfig: context_length must be greater than zero" );
            context_length_ = n;
            return *this;
        }
        GemmaModelConfig& withWeightQuantization( WeightQuantization wq ) { weight_quantization_ = wq; return *this; }
        GemmaModelConfig& withKvCacheCompression( KvCacheCompression kv ) { kv_cache_compression_ = kv; return *this; }
        /// MI355X deployment knobs (no reference counterpart: a 12 GB card always chunks and always bounds the ring)
        GemmaModelConfig& withPrefillChunk( dim_t chunk ) { prefill_chunk_ = chunk; return *this; }          ///< 0 = min(context, 2048)
        GemmaModelConfig& withBoundedLocalKv( bool on ) { bounded_local_kv_ = on; return *this; }            ///< SlidingWindowKvCache on the sliding-window layers
        dim_t getContextLength() const noexcept { return context_length_; }
        WeightQuantization getWeightQuantization() const noexcept { return weight_quantization_; }
        KvCacheCompression getKvCacheCompression() const noexcept { return kv_cache_compression_; }
        dim_t getPrefillChunk() const noexcept { return prefill_chunk_ > 0 ? std::min( prefill_chunk_, context_length_ ) : std::min<dim_t>( context_length_, 2048 ); }
        bool boundedLocalKv() const noexcept { return bounded_local_kv_; }
    private:
        dim_t context_length_{ 0 }, prefill_chunk_{ 0 };
        WeightQuantization weight_quantization_{ WeightQuantization::None };
        KvCacheCompression kv_cache_compression_{ KvCacheCompression::None };
        bool bounded_local_kv_{ false };
    };

    /// Models/QuantizationDispatch.ixx: the runtime deployment choice picks the compile-time policy
    template<TensorDataType TPrecision, typename TResult, typename TAction>
    TResult dispatchWeightQuantization( WeightQuantization wq, KvCacheCompression kv, std::string_view caller, TAction&& action )
    {
        static_assert( TPrecision == TensorDataType::BF16, "the CDNA4 backend computes in BF16" );
        switch ( wq )
        {
            case WeightQuantization::FP4: return action.template operator()<Quant::Weight::PerGroupFp4<128>>();
            case WeightQuantization::FP8: return action.template operator()<Quant::Weight::PerChannelFp8<>>();
            case WeightQuantization::None:
            default:
                if ( kv == KvCacheCompression::FP8 ) throw std::runtime_error( std::string( caller ) + ": FP8 KV cache compression is not yet supported" );
                return action.template operator()<Quant::Weight::NoWeightQuant>();
        }
    }

    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class GemmaModel
    {
        static_assert( TDeviceType == DeviceType::Rocm && TPrecision == TensorDataType::BF16, "GemmaModel<Rocm, BF16>" );
    public:
        template<typename P> using Net = GemmaTransformer<P>;
        using Network = std::variant<std::unique_ptr<Net<Quant::Weight::NoWeightQuant>>, std::unique_ptr<Net<Quant::Weight::PerChannelFp8<>>>,
                                     std::unique_ptr<Net<Quant::Weight::PerGroupFp4<128>>>>;
        using TokenTensor = Tensor<TensorDataType::INT32, Compute::RocmDeviceMemoryResource>;
        static constexpr int32_t kEosToken = 1, kEndOfTurnToken = 106;

        GemmaModel( const GemmaModel& ) = delete;
        GemmaModel& operator=( const GemmaModel& ) = delete;
        ~GemmaModel()
        {
            if ( ring_host_ ) (void)hipHostFree( ring_host_ );
        }

        /// every architectural parameter comes from the artifact's metadata; `model_config` carries the deployment decisions
        static std::unique_ptr<GemmaModel> fromPretrained( const std::string& path, const GemmaModelConfig& model_config, DeviceId device_id = Compute::Device::Rocm( 0 ) )
        {
            if ( device_id.type != TDeviceType ) throw std::invalid_argument( "GemmaModel::fromPretrained: device type mismatch" );
            if ( model_config.getContextLength() == 0 ) throw std::invalid_argument( "GemmaModel::fromPretrained: context_length must be greater than zero" );
            return dispatchWeightQuantization<TPrecision, std::unique_ptr<GemmaModel>>( model_config.getWeightQuantization(), model_config.getKvCacheCompression(), "GemmaModel::fromPretrained",
                [&]<typename TWeightQuantization>()
                {
                    Serialization::PretrainedModelReader reader( path );
                    const auto& md = reader.getPretrainedMetadata();
                    const std::string& artifact = reader.getWeightQuantization();
                    const std::string requested = weightQuantizationName( model_config.getWeightQuantization() );
                    if ( !artifact.empty() && artifact != requested )
                        throw std::runtime_error( "GemmaModel::fromPretrained: artifact '" + path + "' is pre-quantized as '" + artifact + "' but this load requested '" + requested + "'" );
                    if ( reader.metadataJSON().empty() ) throw std::runtime_error( "GemmaModel::fromPretrained: artifact '" + path + "' carries no mila_config metadata" );
                    GemmaConfig cfg = configFromMetadata( md );
                    if ( md.max_seq_length != 0 && model_config.getContextLength() > static_cast<dim_t>( md.max_seq_length ) )
                        throw std::invalid_argument( "GemmaModel::fromPretrained: context_length " + std::to_string( model_config.getContextLength() ) + " exceeds trained max_seq_len " + std::to_string( md.max_seq_length ) );
                    cfg.bounded_local_kv = model_config.boundedLocalKv();
                    auto net = std::make_unique<Net<TWeightQuantization>>( cfg, model_config.getContextLength(), model_config.getPrefillChunk(), device_id );
                    net->loadPretrained( path );
                    return std::unique_ptr<GemmaModel>( new GemmaModel( Network( std::move( net ) ), cfg, model_config, md ) );
                } );
        }
        /// the same model over synthetic parameters (no artifact offline): tests and benchmarks
        static std::unique_ptr<GemmaModel> fromSynthetic( const GemmaConfig& network_config, const GemmaModelConfig& model_config, uint64_t seed,
                                                          const typename Net<Quant::Weight::NoWeightQuant>::SyntheticProfile& profile = {}, DeviceId device_id = Compute::Device::Rocm( 0 ) )
        {
            return dispatchWeightQuantization<TPrecision, std::unique_ptr<GemmaModel>>( model_config.getWeightQuantization(), model_config.getKvCacheCompression(), "GemmaModel::fromSynthetic",
                [&]<typename TWeightQuantization>()
                {
                    GemmaConfig cfg = network_config;
                    cfg.bounded_local_kv = model_config.boundedLocalKv();
                    auto net = std::make_unique<Net<TWeightQuantization>>( cfg, model_config.getContextLength(), model_config.getPrefillChunk(), device_id );
                    typename Net<TWeightQuantization>::SyntheticProfile p;
                    p.linear_gain = profile.linear_gain; p.qk_norm_center = profile.qk_norm_center; p.post_norm_center = profile.post_norm_center; p.layer_scalar = profile.layer_scalar; p.table_gain = profile.table_gain;
                    net->initSynthetic( seed, p );
                    return std::unique_ptr<GemmaModel>( new GemmaModel( Network( std::move( net ) ), cfg, model_config, Serialization::PretrainedMetadata{} ) );
                } );
        }
        static GemmaConfig configFromMetadata( const Serialization::PretrainedMetadata& md )
        {
            GemmaConfig c;
            auto take = [&]( dim_t& dst, uint32_t v ) { if ( v != 0 ) dst = static_cast<dim_t>( v ); };
            take( c.vocab_size, md.vocab_size ); take( c.embedding_dim, md.embedding_dim ); take( c.num_layers, md.num_layers ); take( c.num_heads, md.num_heads );
            take( c.num_kv_heads, md.num_kv_heads ); take( c.head_dim, md.head_dim ); take( c.hidden_dim, md.hidden_dim ); take( c.global_head_dim, md.global_head_dim );
            take( c.num_global_kv_heads, md.num_global_kv_heads ); take( c.window, md.window ); take( c.sliding_window_pattern, md.sliding_window_pattern );
            take( c.global_rotary_dim, md.global_rotary_dim );
            if ( md.norm_epsilon > 0.0f ) c.rms_norm_eps = md.norm_epsilon;
            if ( md.rope_theta_local > 0.0f ) c.rope_theta_local = md.rope_theta_local;
            if ( md.rope_theta_global > 0.0f ) c.rope_theta_global = md.rope_theta_global;
            c.final_logit_softcapping = md.final_logit_softcapping;
            c.validate();
            return c;
        }

        const GemmaConfig& getNetworkConfig() const noexcept { return network_config_; }
        const GemmaModelConfig& getModelConfig() const noexcept { return model_config_; }
        dim_t contextLength() const noexcept { return model_config_.getContextLength(); }
        dim_t vocabSize() const noexcept { return network_config_.vocab_size; }
        int32_t eosToken() const noexcept { return kEosToken; }
        std::unordered_set<int32_t> stopTokens() const { return { kEosToken, kEndOfTurnToken }; }
        float finalLogitSoftcap() const noexcept { return network_config_.final_logit_softcapping; }
        void seedSampler( uint64_t seed ) { rng_.seed( seed ); }
        /// tokens the KV caches currently hold (prompt + everything decoded into them), the key of the prefix reuse
        const std::vector<int32_t>& kvTokenHistory() const noexcept { return kv_token_history_; }
        /// prompt tokens the last generate() did NOT have to prefill
        dim_t lastReusedPrefix() const noexcept { return last_reuse_; }
        Network& network() noexcept { return network_; }

        /// prefill + decode; on_token is called for every generated token except a stop token
        [[nodiscard]] GenerateStatus generate( std::span<const int32_t> prompt_tokens, const std::function<void( int32_t )>& on_token, const GenerateParams& params = {},
                                               const std::atomic<bool>* stop = nullptr )
        {
            return std::visit( [&]( auto& net )
            {
                try { return onGenerating( *net, prompt_tokens, on_token, params, stop ); }
                catch ( const std::invalid_argument& ) { throw; }      // the request checks: nothing was enqueued, the caches are untouched
                catch ( ... )
                {
                    // a replay may still be in flight (awaitSampledToken's timeout, a throwing on_token): drain it before the caller sees the exception,
                    // and drop the reuse key -- what the in-flight step wrote is not in the history
                    try { net->context()->synchronize(); } catch ( ... ) {}
                    kv_token_history_.clear();
                    throw;
                }
            }, network_ );
        }

    private:
        GemmaModel( Network net, const GemmaConfig& cfg, const GemmaModelConfig& mc, Serialization::PretrainedMetadata md )
            : network_( std::move( net ) ), network_config_( cfg ), model_config_( mc ), source_metadata_( std::move( md ) )
        {
            const auto dev = std::visit( []( auto& n ) { return n->context()->getDeviceId(); }, network_ );
            prompt_dev_ = std::make_unique<TokenTensor>( dev, shape_t{ 1, mc.getPrefillChunk() } );
            decode_token_device_ = std::make_unique<TokenTensor>( dev, shape_t{ 1 } );
            seq_dev_ = std::make_unique<Tensor<TensorDataType::INT32, Compute::RocmDeviceMemoryResource>>( dev, shape_t{ 2 } );      // one 64-bit counter
            auto* ctx = std::visit( []( auto& n ) { return n->context(); }, network_ );
            Compute::rocmCheck( mila_cdna4_memset_zero( seq_dev_->rawData(), 8, ctx->getStream() ) );
            hipCheck( hipHostMalloc( reinterpret_cast<void**>( &ring_host_ ), kSnapshots * sizeof( unsigned long long ), hipHostMallocMapped | hipHostMallocCoherent ), "hipHostMalloc (mapped, coherent token ring)" );      // coherence stated, not left to HIP_HOST_COHERENT: the host polls what the device stores
            std::memset( ring_host_, 0, kSnapshots * sizeof( unsigned long long ) );
            void* dptr = nullptr;
            hipCheck( hipHostGetDevicePointer( &dptr, ring_host_, 0 ), "hipHostGetDevicePointer" );
            ring_dev_ = static_cast<unsigned long long*>( dptr );
            ctx->synchronize();
        }

        static bool isGreedy( const SamplingParams& sp ) noexcept { return sp.top_k == 1 || sp.temperature <= 0.0f; }

        // Every sample -- the eager sampler's or the captured step's -- takes the next sequence number on the device and lands in ring slot seq % kSnapshots of
        // host-visible memory.  At most two samples are ever in flight, so a slot is read long before its reuse; the tag makes a stale slot unmistakable.
        unsigned long long* seqCounter() { return reinterpret_cast<unsigned long long*>( seq_dev_->rawData() ); }
        /// sample from the network's current logits into decode_token_device_ (ready for the next decode) and publish it for the host
        template<typename TNet> void enqueueSampleNext( TNet& net, const SamplingParams& sp )
        {
            if ( isGreedy( sp ) ) net.sampleGreedy( *decode_token_device_ );
            else
            {
                typename TNet::SamplingParams p; p.temperature = sp.temperature; p.top_k = sp.top_k; p.top_p = sp.top_p;
                net.sampleStochastic( *decode_token_device_, p, std::uniform_real_distribution<float>( 0.0f, 1.0f )( rng_ ) );
            }
            Compute::rocmCheck( mila_cdna4_snapshot_token( decode_token_device_->data(), seqCounter(), ring_dev_, static_cast<int>( kSnapshots ), net.context()->getStream() ) );
            ++published_;
        }
        /// blocks until sample number `seq` (1-based over this model's lifetime) is host-visible; device work enqueued after it -- the ahead-decoded step --
        /// keeps running.  Polls host memory only; gives up after ~20 s (a wedged device), so a caller never spins forever
        int32_t awaitSampledToken( uint64_t seq )
        {
            volatile unsigned long long* slot = ring_host_ + ( seq % kSnapshots );
            const auto t0 = std::chrono::steady_clock::now();
            for ( uint64_t spins = 0;; ++spins )
            {
                const unsigned long long v = __atomic_load_n( slot, __ATOMIC_ACQUIRE );
                if ( ( v >> 32 ) == ( seq & 0xffffffffull ) ) return static_cast<int32_t>( static_cast<uint32_t>( v ) );
                if ( ( spins & 1023 ) == 1023 )
                {
                    if ( std::chrono::steady_clock::now() - t0 > std::chrono::seconds( 20 ) ) throw std::runtime_error( "GemmaModel::awaitSampledToken: the device did not publish a token within 20 s" );
                    std::this_thread::yield();
                }
            }
        }

        template<typename TNet>
        GenerateStatus onGenerating( TNet& net, std::span<const int32_t> prompt, const std::function<void( int32_t )>& on_token, const GenerateParams& params, const std::atomic<bool>* stop )
        {
            Compute::TraceRange tr( "GemmaModel.generate" );
            if ( prompt.empty() ) throw std::invalid_argument( "GemmaModel::onGenerating: empty prompt" );
            if ( prompt.size() > static_cast<size_t>( contextLength() ) )
                throw std::invalid_argument( "GemmaModel::onGenerating: prompt length " + std::to_string( prompt.size() ) + " exceeds deployment context length " + std::to_string( contextLength() ) );
            for ( int32_t t : prompt )
                if ( t < 0 || t >= vocabSize() ) throw std::invalid_argument( "GemmaModel::onGenerating: token id " + std::to_string( t ) + " outside the vocabulary" );
            std::unordered_set<int32_t> stop_ids;
            if ( params.stop_tokens.empty() ) stop_ids = stopTokens();
            else stop_ids.insert( params.stop_tokens.begin(), params.stop_tokens.end() );

            // transparent KV prefix reuse: cache positions [0, n) are a function of the first n tokens only, so token equality against what the
            // caches hold is the whole validity test; at least the last prompt position is always prefilled (fresh logits)
            const dim_t seq_len = static_cast<dim_t>( prompt.size() );
            dim_t common = 0;
            const dim_t comparable = std::min<dim_t>( seq_len, static_cast<dim_t>( kv_token_history_.size() ) );
            while ( common < comparable && kv_token_history_[ static_cast<size_t>( common ) ] == prompt[ static_cast<size_t>( common ) ] ) ++common;
            dim_t reuse = std::min( common, seq_len - 1 );
            if ( reuse > 0 && !net.rewindKvCache( reuse, static_cast<dim_t>( kv_token_history_.size() ) ) ) reuse = 0;      // a refused rewind (ring staleness) falls back to the full prefill
            last_reuse_ = reuse;
            const dim_t chunk = model_config_.getPrefillChunk();
            auto* ctx = net.context();
            // the history claims only what the caches hold at every instant: cut to the reused prefix before the first chunk overwrites positions >= reuse,
            // extended chunk by chunk, so a chunk that throws leaves no stale claim for the next generate() to "reuse"
            kv_token_history_.resize( static_cast<size_t>( reuse ) );
            for ( dim_t p0 = reuse; p0 < seq_len; p0 += chunk )
            {
                const dim_t n = std::min( chunk, seq_len - p0 );
                Compute::rocmCheck( mila_cdna4_memcpy_h2d( prompt_dev_->rawData(), prompt.data() + p0, static_cast<size_t>( n ) * 4, ctx->getStream() ) );
                net.prefill( *prompt_dev_, n, p0 );
                ctx->synchronize();       // prompt_dev_ is reused by the next chunk
                kv_token_history_.insert( kv_token_history_.end(), prompt.begin() + p0, prompt.begin() + p0 + n );
            }

            const bool greedy = isGreedy( params.sampling );
            dim_t position = seq_len;
            enqueueSampleNext( net, params.sampling );
            int emitted = 0;
            const int max_new = params.max_new_tokens.value_or( static_cast<int>( contextLength() ) );
            if ( position < contextLength() )
            {
                // greedy: the captured step ends with the sampler and the publish; stochastic: the captured step ends at the logits and the sampler (its uniform
                // draw is a host scalar) follows eagerly.  Switching between the two kinds of request re-captures (a few ms, once per switch)
                net.setSampleInGraph( greedy );
                net.setTokenRing( ring_dev_, static_cast<int>( kSnapshots ), seqCounter() );
                net.ensureGraph( *decode_token_device_, position );
                net.setDevicePosition( position );
            }
            while ( true )
            {
                if ( stop && stop->load( std::memory_order_relaxed ) ) { ctx->synchronize(); return GenerateStatus::ClientCancelled; }
                // decode ahead only when another step could consume its logits: within the token budget and with KV-cache room
                const bool more_steps_allowed = emitted + 1 < max_new;
                const bool cache_has_room = position < contextLength();
                const bool ahead = more_steps_allowed && cache_has_room;
                const uint64_t mine = published_;          // the sample this iteration reports
                if ( ahead )
                {
                    net.ensureGraph( *decode_token_device_, position );      // (re-captures only when the position leaves the captured live-length bucket)
                    net.replayGraph();
                    if ( greedy ) ++published_;      // the captured step ends with sampler + publish: the NEXT token
                }
                const int32_t token = awaitSampledToken( mine );
                if ( ahead ) { kv_token_history_.push_back( token ); ++position; }     // the ahead-decode entered it into the caches, whatever it is
                if ( stop_ids.contains( token ) ) { ctx->synchronize(); return GenerateStatus::Success; }
                on_token( token );
                ++emitted;
                if ( !ahead ) return more_steps_allowed ? GenerateStatus::ContextOverflow : GenerateStatus::MaxNewTokensReached;
                if ( !greedy ) enqueueSampleNext( net, params.sampling );
            }
        }

        static constexpr size_t kSnapshots = 8;
        Network network_;
        GemmaConfig network_config_;
        GemmaModelConfig model_config_;
        Serialization::PretrainedMetadata source_metadata_;
        std::unique_ptr<TokenTensor> prompt_dev_, decode_token_device_, seq_dev_;
        std::vector<int32_t> kv_token_history_;
        dim_t last_reuse_{ 0 };
        std::mt19937_64 rng_{ 0x4d494c41ull };
        unsigned long long* ring_host_{ nullptr };      // pinned + mapped: the device writes, the host polls
        unsigned long long* ring_dev_{ nullptr };       // the same memory through its device address
        uint64_t published_{ 0 };                       // samples enqueued so far (their sequence numbers are 1 .. published_)
    };
}
