// The remaining leaf components of the Gemma forward() hot path and the block that wires them, mirroring the reference's public
// surface so that code written against it compiles unchanged against DeviceType::Rocm:
//   Rope                   Components/Encodings/Rope/Rope.ixx:99-200, Rope.Config.ixx:51-130
//   GroupedQueryAttention  Components/Attention/GQA/GroupedQueryAttention.ixx:108-400, GroupedQueryAttention.Config.ixx:59-200
//   Swiglu<.., Gelu>       Components/FFN/Swiglu/Swiglu.ixx:92 (GeGLU: Gemma.Block.ixx:150)
//   Residual               Components/Connections/Residual.ixx:93-127
//   TokenEmbedding         Components/Embeddings/TokenEmbedding.ixx:155-190, 336-384 (raw table shared with the tied lm_head)
//   IDecoderLayer          Components/Transformers/Gemma/IDecoderLayer.ixx:53-76
//   GemmaBlock<kGlobal>    Components/Transformers/Gemma/Gemma.Block.ixx:197-356 (prefill / decode), :858-921 (the child graph and its names)
// Each component resolves, through OperationTraits, to the CDNA4 op classes of Operations.h; none of them computes on the host.
#pragma once

#include <cstring>

#include "Components.h"

namespace Mila::Dnn
{
    enum class ActivationType { Silu, Gelu };

    // ---------------------------------------------------------------------------------------
    // Rope
    // ---------------------------------------------------------------------------------------
    class RopeConfig
    {
    public:
        RopeConfig( dim_t channels, dim_t n_heads, dim_t n_kv_heads, dim_t max_seq_len )
            : channels_( channels ), n_heads_( n_heads ), n_kv_heads_( n_kv_heads ), max_seq_( max_seq_len ) {}
        template<typename Self> decltype( auto ) withBase( this Self&& s, float base ) { s.base_ = base; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withRotaryDim( this Self&& s, dim_t r ) { s.rotary_dim_ = r; return std::forward<Self>( s ); }
        dim_t getEmbeddingDim() const noexcept { return channels_; }
        dim_t getNumHeads() const noexcept { return n_heads_; }
        dim_t getNumKVHeads() const noexcept { return n_kv_heads_; }
        dim_t getHeadDim() const noexcept { return n_heads_ > 0 ? channels_ / n_heads_ : 0; }
        dim_t getMaxSequenceLength() const noexcept { return max_seq_; }
        dim_t getRotaryDim() const noexcept { return rotary_dim_; }
        float getBase() const noexcept { return base_; }
        void validate() const
        {
            if ( channels_ <= 0 || n_heads_ <= 0 || n_kv_heads_ <= 0 || max_seq_ <= 0 ) throw std::invalid_argument( "RopeConfig: dimensions must be positive" );
            if ( channels_ % n_heads_ != 0 ) throw std::invalid_argument( "RopeConfig: channels must be a multiple of n_heads" );
            if ( n_heads_ % n_kv_heads_ != 0 ) throw std::invalid_argument( "RopeConfig: n_heads must be a multiple of n_kv_heads" );
            if ( getHeadDim() % 2 != 0 ) throw std::invalid_argument( "RopeConfig: head_dim must be even" );
            if ( rotary_dim_ < 0 || rotary_dim_ > getHeadDim() || rotary_dim_ % 2 != 0 ) throw std::invalid_argument( "RopeConfig: rotary_dim must be even and within head_dim" );
            if ( !( base_ > 0.0f ) ) throw std::invalid_argument( "RopeConfig: base must be positive" );
        }
    private:
        dim_t channels_, n_heads_, n_kv_heads_, max_seq_;
        float base_{ 10000.0f };
        dim_t rotary_dim_{ 0 };      ///< 0 = the whole head
    };

    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class Rope : public Component<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using OpType = typename OperationTraits<OperationType::RopeOp, TDeviceType, TPrecision>::type;

        Rope( const std::string& name, const RopeConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }

        /// rotate Q [B,T,NH*HS] and K [B,T,NKV*HS] in place for positions position_offset .. position_offset + T - 1
        void prefill( TensorType& Q, TensorType& K, dim_t position_offset )
        {
            requireBuilt( "prefill" );
            const auto& s = Q.shape();
            if ( s.size() != 3 || K.shape().size() != 3 || K.shape()[ 0 ] != s[ 0 ] || K.shape()[ 1 ] != s[ 1 ] )
                throw std::invalid_argument( this->getName() + ": Q and K must be [B, T, heads * head_dim] with equal B and T" );
            if ( s[ 2 ] != config_.getNumHeads() * config_.getHeadDim() || K.shape()[ 2 ] != config_.getNumKVHeads() * config_.getHeadDim() )
                throw std::invalid_argument( this->getName() + ": Q / K widths do not match the configured heads" );
            if ( position_offset < 0 || position_offset + s[ 1 ] > config_.getMaxSequenceLength() )
                throw std::invalid_argument( this->getName() + ": positions beyond the built cache" );
            operation_->prefill( Q, K, static_cast<int>( s[ 0 ] ), static_cast<int>( s[ 1 ] ), static_cast<int>( position_offset ) );
        }
        void decode( TensorType& Q, TensorType& K, dim_t position ) { prefill( Q, K, position ); }
        void forward( TensorType& Q, TensorType& K ) { prefill( Q, K, 0 ); }

        const float* cosCache() const noexcept { return operation_->cosCache(); }
        const float* sinCache() const noexcept { return operation_->sinCache(); }
        const RopeConfig& getConfig() const noexcept { return config_; }
        OpType& getOperation() { return *operation_; }
        /// the cos / sin tables: what one owner pays (they are shared through RopeCacheRegistry; a composite subtracts the duplicates, Gemma.ixx:455-462)
        MemoryStats getMemoryStats() const override { MemoryStats st; if ( operation_ && this->isBuilt() ) st.device_state_bytes = operation_->stateBytes(); return st; }
        MemoryStats getRequiredMemory( const BuildContext& ) const override { MemoryStats st; if ( operation_ ) st.device_state_bytes = operation_->requiredStateBytes(); return st; }

    protected:
        void onExecutionContextSet() override
        {
            operation_ = std::make_shared<OpType>( this->getExecutionContext(),
                Compute::RopeOpConfig{ config_.getMaxSequenceLength(), config_.getHeadDim(), config_.getNumHeads(), config_.getNumKVHeads(), config_.getBase(), config_.getRotaryDim() } );
        }
        void onBuilding( const BuildContext& ctx ) override { operation_->build( ctx ); }
    private:
        void requireBuilt( const char* what ) const { if ( !this->isBuilt() ) throw std::runtime_error( std::string( "Rope must be built before calling " ) + what + "()." ); }
        RopeConfig config_;
        std::shared_ptr<OpType> operation_;
    };

    // ---------------------------------------------------------------------------------------
    // GroupedQueryAttention over an op-owned KV cache
    // ---------------------------------------------------------------------------------------
    class GqaConfig
    {
    public:
        GqaConfig( dim_t model_dim, dim_t num_heads, dim_t num_kv_heads ) : model_dim_( model_dim ), num_heads_( num_heads ), num_kv_heads_( num_kv_heads ) {}
        template<typename Self> decltype( auto ) withModelDim( this Self&& s, dim_t d ) { s.model_dim_ = d; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withNumHeads( this Self&& s, dim_t n ) { s.num_heads_ = n; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withNumKvHeads( this Self&& s, dim_t n ) { s.num_kv_heads_ = n; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withWindow( this Self&& s, dim_t w ) { s.window_ = w; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withAttentionScale( this Self&& s, float a ) { s.scale_ = a; return std::forward<Self>( s ); }
        dim_t getModelDim() const noexcept { return model_dim_; }
        dim_t getNumHeads() const noexcept { return num_heads_; }
        dim_t getNumKvHeads() const noexcept { return num_kv_heads_; }
        dim_t getHeadDim() const noexcept { return num_heads_ > 0 ? model_dim_ / num_heads_ : 0; }
        dim_t getWindow() const noexcept { return window_; }
        float getAttentionScale() const noexcept { return scale_; }
        void validate() const
        {
            if ( model_dim_ <= 0 || num_heads_ <= 0 || num_kv_heads_ <= 0 ) throw std::invalid_argument( "GqaConfig: dimensions must be positive" );
            if ( model_dim_ % num_heads_ != 0 ) throw std::invalid_argument( "GqaConfig: model_dim must be a multiple of num_heads" );
            if ( num_heads_ % num_kv_heads_ != 0 ) throw std::invalid_argument( "GqaConfig: num_heads must be a multiple of num_kv_heads" );
            if ( window_ < 0 ) throw std::invalid_argument( "GqaConfig: window must be >= 0" );
        }
    private:
        dim_t model_dim_, num_heads_, num_kv_heads_;
        dim_t window_{ 0 };          ///< 0 = global
        float scale_{ 0.0f };        ///< <= 0 -> 1 / sqrt(head_dim)
    };

    template<DeviceType TDeviceType, TensorDataType TPrecision, typename TKvPolicy = Quant::KvCache::NoKvCompression>
    class GroupedQueryAttention : public Component<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using OpType = typename OperationTraits<OperationType::GroupedQueryAttentionOp, TDeviceType, TPrecision, TKvPolicy>::type;

        GroupedQueryAttention( const std::string& name, const GqaConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }

        /// chunked prefill: q [B,T,NH*HS], k / v [B,T,NKV*HS] at absolute positions position_offset .. ; appends K/V, returns [B,T,model_dim]
        TensorType& prefill( const TensorType& q, const TensorType& k, const TensorType& v, dim_t position_offset )
        {
            requireBuilt( "prefill" );
            const dim_t B = q.shape()[ 0 ], T = q.shape()[ 1 ];
            checkOperands( q, k, v, B, T );
            if ( T > chunk_ ) throw std::invalid_argument( this->getName() + ": chunk exceeds the built prefill chunk" );
            view_ = std::make_unique<TensorType>( output_->view( shape_t{ B, T, config_.getModelDim() } ) );
            operation_->prefill( q, k, v, *view_, static_cast<int>( T ), static_cast<int>( position_offset ) );
            decode_active_ = false;
            return *view_;
        }
        /// one token per sequence at absolute position `position_offset`
        TensorType& decode( const TensorType& q, const TensorType& k, const TensorType& v, dim_t position_offset )
        {
            requireBuilt( "decode" );
            const dim_t B = q.shape()[ 0 ];
            checkOperands( q, k, v, B, 1 );
            view_ = std::make_unique<TensorType>( output_->view( shape_t{ B, 1, config_.getModelDim() } ) );
            operation_->decode( q, k, v, *view_, static_cast<int>( position_offset ) );
            decode_active_ = true;
            return *view_;
        }
        /// the chunk's K/V rows were already appended by the fused q/k/v post-processing kernel: attention only, into a caller tensor
        void prefillFromCache( const TensorType& q, TensorType& out, int chunk, int position ) { operation_->prefillFromCache( q, out, chunk, position ); }

        bool supportsKVCache() const noexcept { return true; }
        void resetKVCache() { operation_->resetKvCache(); decode_active_ = false; }
        bool rewindKvCache( dim_t position )
        {
            try { operation_->rewindKvCache( position ); }
            catch ( const std::exception& ) { return false; }
            return true;
        }
        void setUseFlashPrefill( bool enabled ) { if ( !enabled ) throw std::invalid_argument( this->getName() + ": the CDNA4 backend has one (flash) prefill path" ); }
        void setUseFlashDecode( bool enabled ) { if ( !enabled ) throw std::invalid_argument( this->getName() + ": the CDNA4 backend has one (flash-decode) path" ); }
        /// all layers of a model may share one attention output (GroupedQueryAttention.ixx:461)
        void installSharedOutput( std::shared_ptr<TensorType> output )
        {
            if ( this->isBuilt() ) throw std::runtime_error( this->getName() + ": installSharedOutput() must precede build()" );
            output_ = std::move( output );
            output_installed_ = true;
        }
        /// state: the op-owned K / V caches (capacity rule CudaGqaOp.ixx:552-574) + the output buffer unless installed
        MemoryStats getMemoryStats() const override
        {
            MemoryStats st;
            if ( operation_ ) st.device_state_bytes += operation_->stateBytes();
            if ( !output_installed_ ) st.device_state_bytes += tensorBytes( output_ );
            return st;
        }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 3 ) throw std::invalid_argument( this->getName() + ": build shape must be [B, max_seq, packed QKV width]" );
            const dim_t chunk = ctx.prefillChunkSize() > 0 ? std::min( ctx.prefillChunkSize(), s[ 1 ] ) : s[ 1 ];
            MemoryStats st;
            st.device_state_bytes += operation_->requiredStateBytes( static_cast<int>( s[ 0 ] ), s[ 1 ], chunk );
            if ( !output_installed_ && !ctx.isOutputInstalled() ) st.device_state_bytes += static_cast<size_t>( s[ 0 ] * chunk * config_.getModelDim() ) * TensorType::kElemBytes;
            return st;
        }

        uint16_t* keyCache() noexcept { return operation_->keyCache(); }
        uint16_t* valueCache() noexcept { return operation_->valueCache(); }
        dim_t cacheCapacity() const noexcept { return operation_->cacheCapacity(); }
        dim_t cacheLength() const noexcept { return operation_->cacheLength(); }
        void noteCacheLength( dim_t length ) { operation_->noteCacheLength( length ); }
        float scale() const noexcept { return operation_->scale(); }
        const GqaConfig& getConfig() const noexcept { return config_; }
        OpType& getOperation() { return *operation_; }

    protected:
        void onExecutionContextSet() override
        {
            operation_ = std::make_shared<OpType>( this->getExecutionContext(),
                Compute::GqaOpConfig{ config_.getNumHeads(), config_.getNumKvHeads(), config_.getHeadDim(), config_.getWindow(), config_.getAttentionScale() } );
        }
        /// input shape [B, max_seq, packed QKV width]; the context's prefill chunk bounds one prefill() call (0 = max_seq)
        void onBuilding( const BuildContext& ctx ) override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 3 ) throw std::invalid_argument( this->getName() + ": build shape must be [B, max_seq, packed QKV width]" );
            const dim_t want = ( config_.getNumHeads() + 2 * config_.getNumKvHeads() ) * config_.getHeadDim(), kv_shared = ( config_.getNumHeads() + config_.getNumKvHeads() ) * config_.getHeadDim();
            if ( s[ 2 ] != want && s[ 2 ] != kv_shared ) throw std::invalid_argument( this->getName() + ": packed QKV width " + std::to_string( s[ 2 ] ) + " matches neither [Q|K|V] nor [Q|K]" );
            chunk_ = ctx.prefillChunkSize() > 0 ? std::min( ctx.prefillChunkSize(), s[ 1 ] ) : s[ 1 ];
            operation_->initializeKvCache( static_cast<int>( s[ 0 ] ), s[ 1 ], chunk_ );
            if ( !output_ ) output_ = std::make_shared<TensorType>( this->getExecutionContext()->getDeviceId(), shape_t{ s[ 0 ], chunk_, config_.getModelDim() } );
            else if ( output_->size() < static_cast<size_t>( s[ 0 ] * chunk_ * config_.getModelDim() ) ) throw std::invalid_argument( this->getName() + ": the installed output is too small" );
        }
    private:
        void requireBuilt( const char* what ) const { if ( !this->isBuilt() ) throw std::runtime_error( std::string( "GroupedQueryAttention must be built before calling " ) + what + "()." ); }
        void checkOperands( const TensorType& q, const TensorType& k, const TensorType& v, dim_t B, dim_t T ) const
        {
            const dim_t qw = config_.getModelDim(), kw = config_.getNumKvHeads() * config_.getHeadDim();
            if ( q.size() != static_cast<size_t>( B * T * qw ) || k.size() != static_cast<size_t>( B * T * kw ) || v.size() != static_cast<size_t>( B * T * kw ) )
                throw std::invalid_argument( this->getName() + ": q / k / v sizes do not match [B, T, heads * head_dim]" );
        }
        GqaConfig config_;
        std::shared_ptr<OpType> operation_;
        std::shared_ptr<TensorType> output_;
        std::unique_ptr<TensorType> view_;
        dim_t chunk_{ 0 };
        bool decode_active_{ false }, output_installed_{ false };
    };

    // ---------------------------------------------------------------------------------------
    // Swiglu (gated activation over a fused [gate | up] input) and Residual
    // ---------------------------------------------------------------------------------------
    class SwigluConfig
    {
    public:
        SwigluConfig() = default;
        void validate() const {}
    };

    template<DeviceType TDeviceType, TensorDataType TPrecision, ActivationType TActivation = ActivationType::Silu>
    class Swiglu : public Component<TDeviceType, TPrecision>
    {
        static_assert( TActivation == ActivationType::Gelu, "the CDNA4 backend implements the Gelu gate (GeGLU, Gemma); the Silu gate is not on the forward() path built here" );
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using OpType = typename OperationTraits<OperationType::GegluOp, TDeviceType, TPrecision>::type;
        Swiglu( const std::string& name, const SwigluConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) {}

        /// input [.., 2H] = [gate | up] -> gelu_tanh(gate) * up, [.., H]
        TensorType& forward( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Swiglu must be built before calling forward()." );
            auto s = input.shape();
            if ( s.back() % 2 != 0 ) throw std::invalid_argument( this->getName() + ": the last dimension must be even ([gate | up])" );
            s.back() /= 2;
            if ( shapeSize( s ) > output_->size() ) throw std::invalid_argument( this->getName() + ": input exceeds the built shape" );
            view_ = std::make_unique<TensorType>( output_->view( s ) );
            operation_->forward( input, *view_ );
            return *view_;
        }
        void installSharedOutput( std::shared_ptr<TensorType> output ) { output_ = std::move( output ); output_installed_ = true; }
        MemoryStats getMemoryStats() const override { MemoryStats st; if ( !output_installed_ ) st.device_state_bytes = tensorBytes( output_ ); return st; }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            MemoryStats st;
            if ( !output_installed_ && !ctx.isOutputInstalled() ) st.device_state_bytes = static_cast<size_t>( shapeSize( ctx.inputShape() ) / 2 ) * TensorType::kElemBytes;
            return st;
        }
    protected:
        void onExecutionContextSet() override { operation_ = std::make_shared<OpType>( this->getExecutionContext() ); }
        void onBuilding( const BuildContext& ctx ) override
        {
            auto s = ctx.inputShape();
            if ( s.back() % 2 != 0 ) throw std::invalid_argument( this->getName() + ": the last dimension must be even ([gate | up])" );
            s.back() /= 2;
            if ( !output_ ) output_ = std::make_shared<TensorType>( this->getExecutionContext()->getDeviceId(), s );
        }
    private:
        SwigluConfig config_;
        std::shared_ptr<OpType> operation_;
        std::shared_ptr<TensorType> output_;
        std::unique_ptr<TensorType> view_;
        bool output_installed_{ false };
    };

    class ResidualConfig
    {
    public:
        ResidualConfig() = default;
        template<typename Self> decltype( auto ) withScalingFactor( this Self&& s, float f ) { s.scaling_ = f; return std::forward<Self>( s ); }
        float getScalingFactor() const noexcept { return scaling_; }
        void validate() const
        {
            if ( !( scaling_ > 0.0f ) ) throw std::invalid_argument( "ResidualConfig: scaling_factor must be > 0" );
            if ( scaling_ != 1.0f ) throw std::invalid_argument( "ResidualConfig: the CDNA4 backend implements the plain sum (scaling_factor 1), the only form on the forward() path" );
        }
    private:
        float scaling_{ 1.0f };
    };

    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class Residual : public Component<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using OpType = typename OperationTraits<OperationType::ResidualOp, TDeviceType, TPrecision>::type;
        Residual( const std::string& name, const ResidualConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }

        TensorType& forward( const TensorType& input_a, const TensorType& input_b )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Residual must be built before calling forward()." );
            if ( input_a.size() != input_b.size() ) throw std::invalid_argument( this->getName() + ": operand sizes differ" );
            if ( input_a.size() > output_->size() ) throw std::invalid_argument( this->getName() + ": input exceeds the built shape" );
            view_ = std::make_unique<TensorType>( output_->view( input_a.shape() ) );
            operation_->forward( input_a, input_b, *view_ );
            return *view_;
        }
        void installSharedOutput( std::shared_ptr<TensorType> output ) { output_ = std::move( output ); output_installed_ = true; }
        MemoryStats getMemoryStats() const override { MemoryStats st; if ( !output_installed_ ) st.device_state_bytes = tensorBytes( output_ ); return st; }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            MemoryStats st;
            if ( !output_installed_ && !ctx.isOutputInstalled() ) st.device_state_bytes = static_cast<size_t>( shapeSize( ctx.inputShape() ) ) * TensorType::kElemBytes;
            return st;
        }
    protected:
        void onExecutionContextSet() override { operation_ = std::make_shared<OpType>( this->getExecutionContext() ); }
        void onBuilding( const BuildContext& ctx ) override
        {
            if ( !output_ ) output_ = std::make_shared<TensorType>( this->getExecutionContext()->getDeviceId(), ctx.inputShape() );
        }
    private:
        ResidualConfig config_;
        std::shared_ptr<OpType> operation_;
        std::shared_ptr<TensorType> output_;
        std::unique_ptr<TensorType> view_;
        bool output_installed_{ false };
    };

    // ---------------------------------------------------------------------------------------
    // TokenEmbedding: owns the raw [vocab, C] table (bf16, or FP8 E4M3 + one scale per row); the tied lm_head adopts it
    // ---------------------------------------------------------------------------------------
    class TokenEmbeddingConfig
    {
    public:
        TokenEmbeddingConfig() = default;
        template<typename Self> decltype( auto ) withVocabSize( this Self&& s, dim_t v ) { s.vocab_ = v; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withEmbeddingDim( this Self&& s, dim_t d ) { s.dim_ = d; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withEmbeddingScale( this Self&& s, float e ) { s.scale_ = e; return std::forward<Self>( s ); }
        dim_t getVocabSize() const { return vocab_; }
        dim_t getEmbeddingDim() const { return dim_; }
        float getEmbeddingScale() const noexcept { return scale_; }
        void validate() const
        {
            if ( vocab_ <= 0 || dim_ <= 0 ) throw std::invalid_argument( "TokenEmbeddingConfig: vocab_size and embedding_dim must be positive" );
        }
    private:
        dim_t vocab_{ 0 }, dim_{ 0 };
        float scale_{ 1.0f };
    };

    template<DeviceType TDeviceType, TensorDataType TIndex, TensorDataType TPrecision, WeightQuantPolicy TTablePolicy = NoWeightQuant>
    class TokenEmbedding : public Component<TDeviceType, TPrecision>
    {
        static_assert( TIndex == TensorDataType::INT32, "token indices are INT32 (TokenEmbedding.ixx)" );
        static_assert( !TTablePolicy::kIsQuantized || TTablePolicy::kPerChannel, "a quantized table carries one scale per vocabulary row (TokenEmbedding.ixx:72-73)" );
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using EmbeddingTensorType = Tensor<TPrecision, MR>;
        using TokenIndexType = Tensor<TIndex, MR>;
        static constexpr bool kIsQuantized = TTablePolicy::kIsQuantized;
        static constexpr TensorDataType kTableDtype = kIsQuantized ? TTablePolicy::kStorageDtype : TPrecision;
        using TableTensorType = Tensor<kTableDtype, MR>;
        using TableScaleTensorType = Tensor<TTablePolicy::kScaleDtype, MR>;
        using QuantizerOp = typename OperationTraits<OperationType::LinearOp, TDeviceType, TPrecision, TTablePolicy>::type;

        TokenEmbedding( const std::string& name, const TokenEmbeddingConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }

        /// output[b, t, :] = wte[X[b, t], :] * embedding_scale; out-of-range ids raise through the context's error flag
        EmbeddingTensorType& forward( const TokenIndexType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "TokenEmbedding must be built before calling forward()." );
            const auto& s = input.shape();
            if ( s.size() != 2 || s[ 0 ] > max_batch_ || s[ 1 ] > max_seq_ )
                throw std::runtime_error( this->getName() + ": input shape " + shapeToString( s ) + " exceeds built max [" + std::to_string( max_batch_ ) + ", " + std::to_string( max_seq_ ) + "]" );
            view_ = std::make_unique<EmbeddingTensorType>( output_->view( shape_t{ s[ 0 ], s[ 1 ], config_.getEmbeddingDim() } ) );
            gather( input.data(), static_cast<int>( s[ 0 ] * s[ 1 ] ), *view_ );
            return *view_;
        }
        /// the same gather into a caller tensor, token ids already on the device (the graph-captured decode step reads its own sampler output)
        void gather( const int32_t* tokens_dev, int n, EmbeddingTensorType& out )
        {
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            const int C = static_cast<int>( config_.getEmbeddingDim() ), V = static_cast<int>( config_.getVocabSize() );
            if constexpr ( !kIsQuantized )
                Compute::rocmCheck( mila_cdna4_embedding_gather_bf16( out.data(), tokens_dev, static_cast<const uint16_t*>( wte_->rawData() ), n, C, V, config_.getEmbeddingScale(),
                                                                      error_flag_->data(), ctx->getStream() ) );
            else
                Compute::rocmCheck( mila_cdna4_embedding_gather_bf16_qfp8( out.data(), tokens_dev, static_cast<const uint8_t*>( wte_->rawData() ), wte_scale_->data(), n, C, V,
                                                                           config_.getEmbeddingScale(), error_flag_->data(), ctx->getStream() ) );
        }
        /// nonzero once any forward saw an id outside [0, vocab): read and cleared (synchronizes the stream)
        int32_t takeError()
        {
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            int32_t e = 0;
            Compute::rocmCheck( mila_cdna4_memcpy_d2h( &e, error_flag_->data(), 4, ctx->getStream() ) );
            ctx->synchronize();
            if ( e != 0 ) { Compute::rocmCheck( mila_cdna4_memset_zero( error_flag_->data(), 4, ctx->getStream() ) ); ctx->synchronize(); }
            return e;
        }
        int32_t* errorFlag() noexcept { return error_flag_->data(); }

        /// "wte": a bf16 [vocab, C] blob (quantized on load for an FP8 table) or a blob already in the table's storage type; "wte_scale": [vocab] F32
        void loadParameter( const std::string& n, const void* blob, size_t bytes ) override
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "TokenEmbedding: build() must precede loadParameter()" );
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            const size_t V = static_cast<size_t>( config_.getVocabSize() ), C = static_cast<size_t>( config_.getEmbeddingDim() );
            if ( n == "wte" )
            {
                if ( bytes == V * C * 2 )
                {
                    if constexpr ( kIsQuantized )
                    {
                        void* staging = ctx->getScratch( bytes );
                        Compute::rocmCheck( mila_cdna4_memcpy_h2d( staging, blob, bytes, ctx->getStream() ) );
                        quantizer_->quantize( static_cast<const uint16_t*>( staging ), *wte_, *wte_scale_ );
                        ctx->synchronize();
                    }
                    else copyToDevice( *wte_, blob, bytes, ctx );
                }
                else if ( kIsQuantized && bytes == wte_->sizeInBytes() ) copyToDevice( *wte_, blob, bytes, ctx );
                else throw std::invalid_argument( this->getName() + ": wte blob has " + std::to_string( bytes ) + " bytes, expected " + std::to_string( V * C * 2 ) );
            }
            else if ( n == "wte_scale" )
            {
                if constexpr ( kIsQuantized )
                {
                    if ( bytes != wte_scale_->sizeInBytes() ) throw std::invalid_argument( this->getName() + ": wte_scale blob size mismatch" );
                    copyToDevice( *wte_scale_, blob, bytes, ctx );
                }
                else throw std::invalid_argument( this->getName() + ": an unquantized table has no wte_scale" );
            }
            else throw std::invalid_argument( this->getName() + ": unknown parameter '" + n + "'" );
        }

        std::shared_ptr<TableTensorType> getWeightTensorShared() const noexcept { return wte_; }
        std::shared_ptr<TableScaleTensorType> getWeightScalesTensorShared() const noexcept requires kIsQuantized { return wte_scale_; }
        const TokenEmbeddingConfig& getConfig() const noexcept { return config_; }
        size_t getParameterBytes() const { return ( wte_ ? wte_->sizeInBytes() : 0 ) + ( wte_scale_ ? wte_scale_->sizeInBytes() : 0 ); }
        /// parameters: the table (+ one scale per row); state: the [max_batch, max_seq, C] output and the 4-byte index-error flag
        MemoryStats getMemoryStats() const override
        {
            MemoryStats st;
            st.device_parameter_bytes = getParameterBytes();
            st.device_state_bytes = tensorBytes( output_ ) + tensorBytes( error_flag_ );
            return st;
        }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 2 ) throw std::invalid_argument( this->getName() + ": build shape must be [B, T] token ids" );
            const size_t V = static_cast<size_t>( config_.getVocabSize() ), C = static_cast<size_t>( config_.getEmbeddingDim() );
            MemoryStats st;
            st.device_parameter_bytes = V * C * TableTensorType::kElemBytes + ( kIsQuantized ? V * TableScaleTensorType::kElemBytes : 0 );
            st.device_state_bytes = static_cast<size_t>( s[ 0 ] * s[ 1 ] ) * C * EmbeddingTensorType::kElemBytes + TokenIndexType::kElemBytes;
            return st;
        }

    protected:
        void onExecutionContextSet() override
        {
            if constexpr ( kIsQuantized )
                quantizer_ = std::make_shared<QuantizerOp>( this->getExecutionContext(), Compute::LinearOpConfig{ config_.getEmbeddingDim(), config_.getVocabSize(), false } );
        }
        /// input shape [max_batch, max_seq] of token ids
        void onBuilding( const BuildContext& ctx ) override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 2 ) throw std::invalid_argument( this->getName() + ": build shape must be [B, T] token ids" );
            max_batch_ = s[ 0 ]; max_seq_ = s[ 1 ];
            const auto dev = this->getExecutionContext()->getDeviceId();
            wte_ = std::make_shared<TableTensorType>( dev, shape_t{ config_.getVocabSize(), config_.getEmbeddingDim() } );
            wte_->setName( this->getName() + ".wte" );
            if constexpr ( kIsQuantized ) wte_scale_ = std::make_shared<TableScaleTensorType>( dev, shape_t{ config_.getVocabSize() } );
            output_ = std::make_shared<EmbeddingTensorType>( dev, shape_t{ max_batch_, max_seq_, config_.getEmbeddingDim() } );
            error_flag_ = std::make_shared<TokenIndexType>( dev, shape_t{ 1 } );
            auto* rctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            Compute::rocmCheck( mila_cdna4_memset_zero( error_flag_->data(), 4, rctx->getStream() ) );
        }
    private:
        TokenEmbeddingConfig config_;
        std::shared_ptr<TableTensorType> wte_;
        std::shared_ptr<TableScaleTensorType> wte_scale_;
        std::shared_ptr<QuantizerOp> quantizer_;
        std::shared_ptr<EmbeddingTensorType> output_;
        std::unique_ptr<EmbeddingTensorType> view_;
        std::shared_ptr<TokenIndexType> error_flag_;
        dim_t max_batch_{ 0 }, max_seq_{ 0 };
    };

    // ---------------------------------------------------------------------------------------
    // Decoder layer interface and the Gemma block
    // ---------------------------------------------------------------------------------------
    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class IDecoderLayer
    {
    public:
        using TensorType = Tensor<TPrecision, typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource>;
        virtual ~IDecoderLayer() = default;
        virtual TensorType& prefill( const TensorType& input, dim_t position_offset ) = 0;
        virtual TensorType& decode( const TensorType& input, dim_t position ) = 0;
        virtual void resetKVCache() = 0;
    };

    /// what a block needs of the model configuration (Gemma.Config.ixx); per-layer values already resolved for its kind
    struct GemmaBlockConfig
    {
        dim_t model_dim{ 0 }, hidden_dim{ 0 }, num_heads{ 0 }, num_kv_heads{ 0 }, head_dim{ 0 };
        dim_t window{ 0 };              ///< 0 on global blocks
        dim_t rotary_dim{ 0 };          ///< 0 = the whole head (local blocks); the proportional share on global blocks
        float rope_theta{ 10000.0f }, rms_norm_eps{ 1e-6f };
        dim_t max_seq{ 0 };
        void validate() const
        {
            if ( model_dim <= 0 || hidden_dim <= 0 || num_heads <= 0 || num_kv_heads <= 0 || head_dim <= 0 || max_seq <= 0 ) throw std::invalid_argument( "GemmaBlockConfig: dimensions must be positive" );
            if ( num_heads % num_kv_heads != 0 ) throw std::invalid_argument( "GemmaBlockConfig: num_heads must be a multiple of num_kv_heads" );
        }
    };

    /// The block's child graph, independent of its kind: what the transformer's fused schedules (Gemma.h) address directly.
    template<DeviceType TDeviceType, TensorDataType TPrecision, WeightQuantPolicy TWeightQuant>
    class GemmaBlockBase : public Component<TDeviceType, TPrecision>, public IDecoderLayer<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using RmsNormType = RmsNorm<TDeviceType, TPrecision>;
        using RopeType = Rope<TDeviceType, TPrecision>;
        using ResidualType = Residual<TDeviceType, TPrecision>;
        using LinearType = Linear<TDeviceType, TPrecision, TWeightQuant>;
        using GeGLUType = Swiglu<TDeviceType, TPrecision, ActivationType::Gelu>;

        const bool global;
        size_t index{ 0 };              ///< position in the model's layer list (set by the owner)
        std::shared_ptr<RmsNormType> input_norm, q_norm, k_norm, v_norm, post_attn_norm, pre_ffn_norm, post_ffn_norm;
        std::shared_ptr<LinearType> qkv_proj, o_proj, fc_gate_up, fc_down;
        std::shared_ptr<RopeType> rope;
        std::shared_ptr<ResidualType> res_1, res_2;
        std::shared_ptr<GeGLUType> geglu;
        float layer_scalar{ 1.0f };      ///< Gemma 4: hidden_states *= layer_scalar at the end of the block (Gemma.Block.ixx:546-560)

        const GemmaBlockConfig& getConfig() const noexcept { return config_; }
        // the block's geometry under the reference's accessor names (Gemma.Block.ixx:160-192)
        bool isGlobal() const noexcept { return global; }
        dim_t headDim() const noexcept { return config_.head_dim; }
        dim_t numKVHeads() const noexcept { return config_.num_kv_heads; }
        bool keyEqualsValue() const noexcept { return global; }          // global blocks have no v_proj: V = v_norm(raw k_proj)
        dim_t window() const noexcept { return config_.window; }
        dim_t kvProjWidth() const noexcept { return kvWidth(); }
        /// the children's names in construction order (Gemma.Block.ixx:858-921)
        std::vector<std::string> childNames() const
        {
            return { input_norm->getName(), q_norm->getName(), k_norm->getName(), v_norm->getName(), post_attn_norm->getName(), pre_ffn_norm->getName(), post_ffn_norm->getName(),
                     qkv_proj->getName(), rope->getName(), this->getName() + ".gqa", o_proj->getName(), res_1->getName(), fc_gate_up->getName(), geglu->getName(), fc_down->getName(),
                     res_2->getName() };
        }
        dim_t qProjWidth() const noexcept { return config_.num_heads * config_.head_dim; }
        dim_t kvWidth() const noexcept { return config_.num_kv_heads * config_.head_dim; }
        dim_t packedQKVWidth() const noexcept { return qProjWidth() + ( global ? 1 : 2 ) * kvWidth(); }     // global: K = V, no v_proj

        // the attention component's cache surface, whatever its KV policy
        virtual uint16_t* keyCache() noexcept = 0;
        virtual uint16_t* valueCache() noexcept = 0;
        virtual dim_t cacheCapacity() const noexcept = 0;
        virtual float attentionScale() const noexcept = 0;
        virtual void prefillFromCache( const TensorType& q, TensorType& out, int chunk, int position ) = 0;
        /// drop everything from `position` on, given that `written` positions were appended (the fused schedules append behind the op's back); false = refused
        virtual bool rewindKvCache( dim_t position, dim_t written ) = 0;

        /// `layer_scalar` ([1] F32); children load through their own components
        void loadParameter( const std::string& n, const void* blob, size_t bytes ) override
        {
            if ( n != "layer_scalar" ) throw std::invalid_argument( this->getName() + ": unknown parameter '" + n + "'" );
            if ( bytes != 4 ) throw std::invalid_argument( this->getName() + ": layer_scalar is one F32" );
            std::memcpy( &layer_scalar, blob, 4 );
        }

    protected:
        GemmaBlockBase( const std::string& name, const GemmaBlockConfig& cfg, bool is_global ) : Component<TDeviceType, TPrecision>( name ), global( is_global ), config_( cfg ) { config_.validate(); }

        template<typename C, typename... A> std::shared_ptr<C> child( const std::string& leaf, A&&... a )
        {
            auto c = std::make_shared<C>( this->getName() + "." + leaf, std::forward<A>( a )... );
            c->setExecutionContext( this->getExecutionContext() );
            return c;
        }
        /// Gemma.Block.ixx:858-921: the same children under the same names
        void createGraph()
        {
            const dim_t D = config_.model_dim, HD = config_.head_dim;
            auto rms = [&]( dim_t dim ) { return RmsNormConfig( dim ).withEpsilon( config_.rms_norm_eps ).withBias( false ); };
            input_norm = child<RmsNormType>( "input_norm", rms( D ) );
            q_norm = child<RmsNormType>( "q_norm", rms( HD ) );
            k_norm = child<RmsNormType>( "k_norm", rms( HD ) );
            v_norm = child<RmsNormType>( "v_norm", rms( HD ) );
            post_attn_norm = child<RmsNormType>( "post_attn_norm", rms( D ) );
            pre_ffn_norm = child<RmsNormType>( "pre_ffn_norm", rms( D ) );
            post_ffn_norm = child<RmsNormType>( "post_ffn_norm", rms( D ) );
            qkv_proj = child<LinearType>( "qkv_proj", LinearConfig( D, packedQKVWidth() ).withBias( false ) );
            rope = child<RopeType>( "rope", RopeConfig( qProjWidth(), config_.num_heads, config_.num_kv_heads, config_.max_seq ).withBase( config_.rope_theta ).withRotaryDim( config_.rotary_dim ) );
            o_proj = child<LinearType>( "o_proj", LinearConfig( qProjWidth(), D ).withBias( false ) );
            res_1 = child<ResidualType>( "res_1", ResidualConfig{} );
            fc_gate_up = child<LinearType>( "fc_gate_up", LinearConfig( D, 2 * config_.hidden_dim ).withBias( false ) );
            geglu = child<GeGLUType>( "geglu", SwigluConfig() );
            fc_down = child<LinearType>( "fc_down", LinearConfig( config_.hidden_dim, D ).withBias( false ) );
            res_2 = child<ResidualType>( "res_2", ResidualConfig{} );
        }
        /// build shape [B, chunk, model_dim]: every child is built for one prefill chunk
        void buildChildren( const BuildContext& ctx )
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 3 || s[ 2 ] != config_.model_dim ) throw std::invalid_argument( this->getName() + ": input must be rank 3 [B, T, model_dim]" );
            const dim_t B = s[ 0 ], P = s[ 1 ], D = config_.model_dim, HD = config_.head_dim, NH = config_.num_heads, NKV = config_.num_kv_heads;
            const auto inf = [&]( const shape_t& shape ) { return BuildContext( shape, RuntimeMode::Inference ); };
            input_norm->build( inf( { B, P, D } ) );
            q_norm->build( inf( { B, P * NH, HD } ) );
            k_norm->build( inf( { B, P * NKV, HD } ) );
            v_norm->build( inf( { B, P * NKV, HD } ) );
            post_attn_norm->build( inf( { B, P, D } ) );
            pre_ffn_norm->build( inf( { B, P, D } ) );
            post_ffn_norm->build( inf( { B, P, D } ) );
            qkv_proj->build( inf( { B, P, D } ) );
            rope->build( inf( { B, P, qProjWidth() } ) );
            o_proj->build( inf( { B, P, qProjWidth() } ) );
            res_1->build( inf( { B, P, D } ) );
            fc_gate_up->build( inf( { B, P, D } ) );
            geglu->build( inf( { B, P, 2 * config_.hidden_dim } ) );
            fc_down->build( inf( { B, P, config_.hidden_dim } ) );
            res_2->build( inf( { B, P, D } ) );
            const auto dev = this->getExecutionContext()->getDeviceId();
            auto need = [&]( std::shared_ptr<TensorType>& t, dim_t width, const char* what )
            {
                if ( !t ) t = std::make_shared<TensorType>( dev, shape_t{ B, P, width } );
                else if ( t->size() < static_cast<size_t>( B * P * width ) ) throw std::invalid_argument( this->getName() + ": the installed " + what + " buffer is too small" );
            };
            need( q_, qProjWidth(), "q" );
            need( k_, kvWidth(), "k" );
            if ( !global ) need( v_, kvWidth(), "v" );
            need( out_, D, "output stream" );
        }
    public:
        /// Blocks run one after the other, so a model may hand all of them the same split scratch (q / k / v) and output stream buffer
        /// (Gemma.Block.ixx:411 installSharedWorkspace); precedes build().  The output may alias the PREVIOUS block's output: a block
        /// reads its input last in res_1 and writes its output last of all.
        void installSharedWorkspace( std::shared_ptr<TensorType> q, std::shared_ptr<TensorType> k, std::shared_ptr<TensorType> v, std::shared_ptr<TensorType> stream )
        {
            if ( this->isBuilt() ) throw std::runtime_error( this->getName() + ": installSharedWorkspace() must precede build()" );
            q_ = std::move( q ); k_ = std::move( k ); if ( !global ) v_ = std::move( v ); out_ = std::move( stream );
            workspace_installed_ = true;
        }
        /// the block's own split scratch / output stream (unless installed: Gemma.Block.ixx:433-440 sums the children, the owner of a pooled workspace counts it once)
        size_t ownWorkspaceBytes() const noexcept { return workspace_installed_ ? 0 : tensorBytes( q_ ) + tensorBytes( k_ ) + tensorBytes( v_ ) + tensorBytes( out_ ); }
        size_t requiredWorkspaceBytes( dim_t B, dim_t P ) const noexcept
        {
            if ( workspace_installed_ ) return 0;
            return static_cast<size_t>( B * P ) * static_cast<size_t>( qProjWidth() + ( global ? 1 : 2 ) * kvWidth() + config_.model_dim ) * TensorType::kElemBytes;
        }
        /// every leaf but the attention component, with the BuildContext buildChildren() gives it
        template<typename F> void forEachLeaf( dim_t B, dim_t P, F&& f ) const
        {
            const dim_t D = config_.model_dim, HD = config_.head_dim, NH = config_.num_heads, NKV = config_.num_kv_heads;
            const auto inf = [&]( const shape_t& shape ) { return BuildContext( shape, RuntimeMode::Inference ); };
            f( *input_norm, inf( { B, P, D } ) ); f( *q_norm, inf( { B, P * NH, HD } ) ); f( *k_norm, inf( { B, P * NKV, HD } ) ); f( *v_norm, inf( { B, P * NKV, HD } ) );
            f( *post_attn_norm, inf( { B, P, D } ) ); f( *pre_ffn_norm, inf( { B, P, D } ) ); f( *post_ffn_norm, inf( { B, P, D } ) );
            f( *qkv_proj, inf( { B, P, D } ) ); f( *rope, inf( { B, P, qProjWidth() } ) ); f( *o_proj, inf( { B, P, qProjWidth() } ) ); f( *res_1, inf( { B, P, D } ) );
            f( *fc_gate_up, inf( { B, P, D } ) ); f( *geglu, inf( { B, P, 2 * config_.hidden_dim } ) ); f( *fc_down, inf( { B, P, config_.hidden_dim } ) ); f( *res_2, inf( { B, P, D } ) );
        }
    protected:
        GemmaBlockConfig config_;
        std::shared_ptr<TensorType> q_, k_, v_, out_;
        bool workspace_installed_{ false };
    };

    /// kGlobal blocks: K = V (no v_proj, V = v_norm(raw k_proj)), the global head_dim / rotary share / theta, an unbounded cache.
    template<DeviceType TDeviceType, TensorDataType TPrecision, bool kGlobal, WeightQuantPolicy TWeightQuant = NoWeightQuant,
             typename TKvPolicy = Quant::KvCache::NoKvCompression>
    class GemmaBlock : public GemmaBlockBase<TDeviceType, TPrecision, TWeightQuant>
    {
    public:
        using Base = GemmaBlockBase<TDeviceType, TPrecision, TWeightQuant>;
        using TensorType = typename Base::TensorType;
        using AttentionType = GroupedQueryAttention<TDeviceType, TPrecision, TKvPolicy>;
        static_assert( !( kGlobal && TKvPolicy::kBoundedRing ), "global blocks attend to the whole context: their cache is unbounded (Gemma.ixx:154)" );

        std::shared_ptr<AttentionType> attn;

        GemmaBlock( const std::string& name, const GemmaBlockConfig& cfg ) : Base( name, cfg, kGlobal ) {}

        TensorType& prefill( const TensorType& input, dim_t position_offset ) override { return run( input, position_offset, false ); }
        TensorType& decode( const TensorType& input, dim_t position ) override
        {
            if ( input.shape().size() != 3 || input.shape()[ 1 ] != 1 ) throw std::invalid_argument( this->getName() + ": decode takes [B, 1, model_dim]" );
            return run( input, position, true );
        }
        void resetKVCache() override { attn->resetKVCache(); }

        uint16_t* keyCache() noexcept override { return attn->keyCache(); }
        uint16_t* valueCache() noexcept override { return attn->valueCache(); }
        dim_t cacheCapacity() const noexcept override { return attn->cacheCapacity(); }
        float attentionScale() const noexcept override { return attn->scale(); }
        void prefillFromCache( const TensorType& q, TensorType& out, int chunk, int position ) override { attn->prefillFromCache( q, out, chunk, position ); }
        bool rewindKvCache( dim_t position, dim_t written ) override { attn->noteCacheLength( written ); return attn->rewindKvCache( position ); }

        /// Gemma.Block.ixx:433-440 / :464-490: the children's stats + the block's own workspace (unless the owner of a pooled one installed it)
        MemoryStats getMemoryStats() const override
        {
            MemoryStats st;
            if ( !this->input_norm ) return st;
            this->forEachLeaf( 1, 1, [&]( const auto& c, const BuildContext& ) { st += c.getMemoryStats(); } );
            st += attn->getMemoryStats();
            st.device_state_bytes += this->ownWorkspaceBytes();
            return st;
        }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 3 || s[ 2 ] != this->config_.model_dim ) throw std::invalid_argument( this->getName() + ": input must be rank 3 [B, T, model_dim]" );
            MemoryStats st;
            this->forEachLeaf( s[ 0 ], s[ 1 ], [&]( const auto& c, const BuildContext& cc ) { st += c.getRequiredMemory( cc ); } );
            st += attn->getRequiredMemory( BuildContext( shape_t{ s[ 0 ], this->config_.max_seq, this->packedQKVWidth() }, RuntimeMode::Inference, false, s[ 1 ] ) );
            st.device_state_bytes += this->requiredWorkspaceBytes( s[ 0 ], s[ 1 ] );
            return st;
        }

    protected:
        void onExecutionContextSet() override
        {
            this->createGraph();
            // head_dim from the Q width, the block's window, scale 1.0: QK-norm controls the magnitude (Gemma.Block.ixx:899-904)
            attn = this->template child<AttentionType>( "gqa", GqaConfig( this->qProjWidth(), this->config_.num_heads, this->config_.num_kv_heads ).withWindow( this->config_.window ).withAttentionScale( 1.0f ) );
        }
        void onBuilding( const BuildContext& ctx ) override
        {
            this->buildChildren( ctx );
            const auto& s = ctx.inputShape();
            attn->build( BuildContext( shape_t{ s[ 0 ], this->config_.max_seq, this->packedQKVWidth() }, RuntimeMode::Inference, false, s[ 1 ] ) );
        }

    private:
        /// the reference's order, one component per step (Gemma.Block.ixx:197-356)
        TensorType& run( const TensorType& input, dim_t position, bool single )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( std::string( "GemmaBlock::" ) + ( single ? "decode" : "prefill" ) + ": must be built before " + ( single ? "decode()." : "prefill()." ) );
            const dim_t B = input.shape()[ 0 ], T = input.shape()[ 1 ];
            const dim_t NH = this->config_.num_heads, NKV = this->config_.num_kv_heads, HD = this->config_.head_dim;
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );

            auto& normed = this->input_norm->forward( input );
            auto& qkv = this->qkv_proj->forward( normed );
            auto q = this->q_->view( shape_t{ B, T, NH * HD } );
            auto k = this->k_->view( shape_t{ B, T, NKV * HD } );
            uint16_t* vp = nullptr;
            if constexpr ( !kGlobal ) vp = this->v_->data();
            Compute::rocmCheck( mila_cdna4_split3_bf16( q.data(), k.data(), vp, static_cast<const uint16_t*>( qkv.rawData() ), static_cast<int>( B * T ), static_cast<int>( NH * HD ),
                                                        static_cast<int>( NKV * HD ), kGlobal ? 0 : static_cast<int>( NKV * HD ), ctx->getStream() ) );
            auto& q_normed = this->q_norm->forward( q.view( shape_t{ B, T * NH, HD } ) );
            auto& k_normed = this->k_norm->forward( k.view( shape_t{ B, T * NKV, HD } ) );
            auto q_roped = q_normed.view( shape_t{ B, T, NH * HD } );
            auto k_roped = k_normed.view( shape_t{ B, T, NKV * HD } );
            if ( single ) this->rope->decode( q_roped, k_roped, position );
            else this->rope->prefill( q_roped, k_roped, position );
            // V is per-head normalized, never rotated; a global block derives it from the RAW key projection, which k_norm (own output) left untouched
            const TensorType v_raw = kGlobal ? k.view( shape_t{ B, T * NKV, HD } ) : this->v_->view( shape_t{ B, T * NKV, HD } );
            auto& v_normed = this->v_norm->forward( v_raw );
            auto v_view = v_normed.view( shape_t{ B, T, NKV * HD } );
            auto& att = single ? attn->decode( q_roped, k_roped, v_view, position ) : attn->prefill( q_roped, k_roped, v_view, position );

            auto& o = this->o_proj->forward( att );
            auto& o_normed = this->post_attn_norm->forward( o );
            auto& res1 = this->res_1->forward( input, o_normed );
            auto& ffn_in = this->pre_ffn_norm->forward( res1 );
            auto& gate_up = this->fc_gate_up->forward( ffn_in );
            auto& ffn_act = this->geglu->forward( gate_up );
            auto& ffn = this->fc_down->forward( ffn_act );
            auto& ffn_normed = this->post_ffn_norm->forward( ffn );
            auto& res2 = this->res_2->forward( res1, ffn_normed );
            view_ = std::make_unique<TensorType>( this->out_->view( shape_t{ B, T, this->config_.model_dim } ) );
            Compute::rocmCheck( mila_cdna4_scale_bf16( view_->data(), res2.data(), static_cast<int64_t>( B * T * this->config_.model_dim ), this->layer_scalar, ctx->getStream() ) );
            return *view_;
        }
        std::unique_ptr<TensorType> view_;
    };
}
