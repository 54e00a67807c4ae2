// The GPT-2 side of the component layer: LayerNorm, Gelu, MultiHeadAttention, Lpe, MLP and GptBlock under the reference's names
//   LayerNorm           Components/Normalization/LayerNorm/LayerNorm.ixx:125-135, LayerNorm.Config.ixx:63-100 (parameters "weight", "bias")
//   Gelu                Components/Activations/Gelu/Gelu.ixx (tanh approximation, ElementwiseActivation.h:41-50)
//   MultiHeadAttention  Components/Attention/MHA/MultiHeadAttention.ixx:124-160, MultiHeadAttention.Config.ixx:42-61
//   Lpe                 Components/Encodings/Lpe/Lpe.ixx:133-156 (parameters "wte", "wpe": :278, 349-358), Lpe.Config.ixx:42-62
//   MLP                 Components/FFN/MLP/MLP.ixx:148-161, children fc_1 / gelu / fc_2 (:410-412)
//   GptBlock            Components/Transformers/Gpt/GptBlock.ixx:148-184, children attn, ln_1, ln_2, fc_qkv_proj, fc_out_proj, res_1, res_2, mlp (:512-546)
// The reference has no BF16 rows for these on CUDA (OPS/OperationTraits.Cuda.ixx:146-149,208-222,234-236,274-282) and its CPU backend is
// FP32-only; the BF16 rows exist for the CDNA4 device and are checked against the FP32 CPU oracle (tests/test_gpt_host_gpu.py).
#pragma once

#include "GemmaBlock.h"

namespace Mila::Dnn
{
    class LayerNormConfig
    {
    public:
        explicit LayerNormConfig( shape_t shape ) : shape_( std::move( shape ) ) {}
        template<typename Self> decltype( auto ) withEpsilon( this Self&& s, float e ) { s.eps_ = e; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withBias( this Self&& s, bool b ) { s.bias_ = b; return std::forward<Self>( s ); }
        const shape_t& getNormalizedShape() const noexcept { return shape_; }
        dim_t dim() const noexcept { return shape_.empty() ? 0 : shape_.back(); }
        float getEpsilon() const noexcept { return eps_; }
        bool hasBias() const noexcept { return bias_; }
        void validate() const
        {
            if ( shape_.size() != 1 || shape_[ 0 ] <= 0 ) throw std::invalid_argument( "LayerNormConfig: the CDNA4 backend normalizes over one trailing dimension of positive size" );
            if ( !( eps_ > 0.0f ) ) throw std::invalid_argument( "LayerNormConfig: epsilon must be positive" );
        }
    private:
        shape_t shape_;
        float eps_{ 1e-5f };
        bool bias_{ true };
    };

    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class LayerNorm : public Component<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using OpType = typename OperationTraits<OperationType::LayerNormOp, TDeviceType, TPrecision>::type;
        LayerNorm( const std::string& name, const LayerNormConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }

        TensorType& forward( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "LayerNorm must be built before calling forward." );
            if ( input.shape().back() != config_.dim() ) throw std::invalid_argument( this->getName() + ": trailing dimension mismatch" );
            if ( input.size() > output_->size() ) throw std::invalid_argument( this->getName() + ": input exceeds the built shape" );
            view_ = std::make_unique<TensorType>( output_->view( input.shape() ) );
            operation_->forward( input, *view_ );
            return *view_;
        }
        void loadParameter( const std::string& n, const void* blob, size_t bytes ) override
        {
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            TensorType* t = n == "weight" ? weight_.get() : ( n == "bias" ? bias_.get() : nullptr );
            if ( !t ) throw std::invalid_argument( this->getName() + ": unknown parameter '" + n + "'" );
            if ( bytes != t->sizeInBytes() ) throw std::invalid_argument( this->getName() + ": " + n + " blob size mismatch" );
            copyToDevice( *t, blob, bytes, ctx );
        }
        const LayerNormConfig& getConfig() const noexcept { return config_; }
        /// LayerNorm.ixx:363: parameters weight (+ bias), state the output buffer (the op keeps no per-row statistics at inference)
        MemoryStats getMemoryStats() const override { MemoryStats st; st.device_parameter_bytes = tensorBytes( weight_ ) + tensorBytes( bias_ ); st.device_state_bytes = tensorBytes( output_ ); return st; }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            MemoryStats st;
            st.device_parameter_bytes = static_cast<size_t>( config_.dim() ) * ( config_.hasBias() ? 2 : 1 ) * TensorType::kElemBytes;
            st.device_state_bytes = static_cast<size_t>( shapeSize( ctx.inputShape() ) ) * TensorType::kElemBytes;
            return st;
        }
    protected:
        void onExecutionContextSet() override
        {
            operation_ = std::make_shared<OpType>( this->getExecutionContext(), Compute::NormOpConfig{ config_.dim(), config_.getEpsilon(), config_.hasBias(), 0.0f } );
        }
        void onBuilding( const BuildContext& ctx ) override
        {
            const auto dev = this->getExecutionContext()->getDeviceId();
            weight_ = std::make_shared<TensorType>( dev, shape_t{ config_.dim() } );
            if ( config_.hasBias() ) bias_ = std::make_shared<TensorType>( dev, shape_t{ config_.dim() } );
            operation_->setParameters( weight_.get(), bias_.get() );
            operation_->build( ctx );
            output_ = std::make_shared<TensorType>( dev, ctx.inputShape() );
        }
    private:
        LayerNormConfig config_;
        std::shared_ptr<OpType> operation_;
        std::shared_ptr<TensorType> weight_, bias_, output_;
        std::unique_ptr<TensorType> view_;
    };

    class GeluConfig { public: GeluConfig() = default; void validate() const {} };

    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class Gelu : public Component<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using OpType = typename OperationTraits<OperationType::GeluOp, TDeviceType, TPrecision>::type;
        Gelu( const std::string& name, const GeluConfig& = GeluConfig() ) : Component<TDeviceType, TPrecision>( name ) {}
        TensorType& forward( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Gelu must be built before calling forward." );
            if ( input.size() > output_->size() ) throw std::invalid_argument( this->getName() + ": input exceeds the built shape" );
            view_ = std::make_unique<TensorType>( output_->view( input.shape() ) );
            operation_->forward( input, *view_ );
            return *view_;
        }
        MemoryStats getMemoryStats() const override { MemoryStats st; st.device_state_bytes = tensorBytes( output_ ); return st; }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override { MemoryStats st; st.device_state_bytes = static_cast<size_t>( shapeSize( ctx.inputShape() ) ) * TensorType::kElemBytes; return st; }
    protected:
        void onExecutionContextSet() override { operation_ = std::make_shared<OpType>( this->getExecutionContext() ); }
        void onBuilding( const BuildContext& ctx ) override { output_ = std::make_shared<TensorType>( this->getExecutionContext()->getDeviceId(), ctx.inputShape() ); }
    private:
        std::shared_ptr<OpType> operation_;
        std::shared_ptr<TensorType> output_;
        std::unique_ptr<TensorType> view_;
    };

    class MultiHeadAttentionConfig
    {
    public:
        MultiHeadAttentionConfig( dim_t model_dim, dim_t num_heads ) : model_dim_( model_dim ), num_heads_( num_heads ) {}
        template<typename Self> decltype( auto ) withModelDim( this Self&& s, dim_t d ) { s.model_dim_ = d; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withNumHeads( this Self&& s, dim_t n ) { s.num_heads_ = n; return std::forward<Self>( s ); }
        dim_t getModelDim() const noexcept { return model_dim_; }
        dim_t getNumHeads() const noexcept { return num_heads_; }
        void validate() const
        {
            if ( model_dim_ <= 0 || num_heads_ <= 0 || model_dim_ % num_heads_ != 0 ) throw std::invalid_argument( "MultiHeadAttentionConfig: model_dim must be a positive multiple of num_heads" );
        }
    private:
        dim_t model_dim_, num_heads_;
    };

    /// causal self-attention over a packed [B, T, 3C] projection (Components/Attention/MHA/MultiHeadAttention.ixx).  forward() is the sole entry point for prefill:
    /// the first call initializes the KV cache and fills it; called again after decode() steps it resets the cache and starts a new session (:124-158).
    /// decode() is the single-token step over the cache (:209-227).
    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class MultiHeadAttention : public Component<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using OpType = typename OperationTraits<OperationType::MultiHeadAttentionOp, TDeviceType, TPrecision>::type;
        MultiHeadAttention( const std::string& name, const MultiHeadAttentionConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }
        TensorType& forward( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "MultiHeadAttention must be built before calling forward()." );
            const auto& s = input.shape();
            if ( s.size() != 3 || s[ 2 ] != 3 * config_.getModelDim() ) throw std::invalid_argument( this->getName() + ": expected a packed [B, T, 3 * model_dim] input" );
            const shape_t os{ s[ 0 ], s[ 1 ], config_.getModelDim() };
            if ( shapeSize( os ) > output_->size() ) throw std::invalid_argument( this->getName() + ": input exceeds the built shape" );
            view_ = std::make_unique<TensorType>( output_->view( os ) );
            if ( s[ 0 ] == max_input_shape_[ 0 ] )
            {
                if ( decode_active_ ) { operation_->resetKvCache(); cache_initialized_ = false; decode_active_ = false; }      // called after decode steps: a new session
                if ( !cache_initialized_ ) { operation_->initializeKvCache( max_input_shape_[ 0 ], max_input_shape_[ 1 ] ); cache_initialized_ = true; }
                operation_->prefill( input, *view_ );      // populates the cache as a side effect
                return *view_;
            }
            operation_->forward( input, *view_ );          // a batch other than the built one has no cache behind it
            return *view_;
        }
        /// Precondition: forward() has populated the cache.  input [B, 1, 3 * model_dim] -> [B, 1, model_dim]
        TensorType& decode( const TensorType& input, dim_t position )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "MultiHeadAttention must be built before calling decode()." );
            if ( !cache_initialized_ ) throw std::runtime_error( this->getName() + ": decode() before forward() has populated the KV cache" );
            operation_->decode( input, *decode_output_, position );
            decode_active_ = true;
            return *decode_output_;
        }
        bool supportsKVCache() const noexcept { return true; }
        void initializeKVCache( dim_t max_seq_len ) { operation_->initializeKvCache( max_input_shape_[ 0 ], max_seq_len ); cache_initialized_ = true; decode_active_ = false; }
        void resetKVCache() { operation_->resetKvCache(); decode_active_ = false; }
        const MultiHeadAttentionConfig& getConfig() const noexcept { return config_; }
        OpType& getOperation() noexcept { return *operation_; }
        /// state: the [B, T, C] output, the [B, 1, C] decode output and -- once initializeKVCache() ran -- the op's K / V caches (CudaMhaOp.ixx:112-143: allocated there, not at build)
        MemoryStats getMemoryStats() const override
        {
            MemoryStats st;
            st.device_state_bytes = tensorBytes( output_ ) + tensorBytes( decode_output_ ) + ( operation_ ? operation_->stateBytes() : 0 );
            return st;
        }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 3 || s[ 2 ] != 3 * config_.getModelDim() ) throw std::invalid_argument( this->getName() + ": build shape must be [B, T, 3 * model_dim]" );
            MemoryStats st;
            st.device_state_bytes = static_cast<size_t>( s[ 0 ] * ( s[ 1 ] + 1 ) * config_.getModelDim() ) * TensorType::kElemBytes;
            return st;
        }
    protected:
        void onExecutionContextSet() override { operation_ = std::make_shared<OpType>( this->getExecutionContext(), config_.getModelDim(), config_.getNumHeads() ); }
        void onBuilding( const BuildContext& ctx ) override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 3 || s[ 2 ] != 3 * config_.getModelDim() ) throw std::invalid_argument( this->getName() + ": build shape must be [B, T, 3 * model_dim]" );
            max_input_shape_ = s;
            const auto dev = this->getExecutionContext()->getDeviceId();
            output_ = std::make_shared<TensorType>( dev, shape_t{ s[ 0 ], s[ 1 ], config_.getModelDim() } );
            decode_output_ = std::make_unique<TensorType>( dev, shape_t{ s[ 0 ], 1, config_.getModelDim() } );
            operation_->build( ctx );
            cache_initialized_ = false; decode_active_ = false;
        }
    private:
        MultiHeadAttentionConfig config_;
        std::shared_ptr<OpType> operation_;
        std::shared_ptr<TensorType> output_;
        std::unique_ptr<TensorType> view_, decode_output_;
        shape_t max_input_shape_;
        bool cache_initialized_{ false }, decode_active_{ false };
    };

    class LpeConfig
    {
    public:
        LpeConfig() = default;
        template<typename Self> decltype( auto ) withEmbeddingDim( this Self&& s, dim_t d ) { s.dim_ = d; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withMaxSequenceLength( this Self&& s, dim_t t ) { s.max_seq_ = t; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withVocabularyLength( this Self&& s, dim_t v ) { s.vocab_ = v; return std::forward<Self>( s ); }
        dim_t getEmbeddingDim() const noexcept { return dim_; }
        dim_t getMaxSequenceLength() const noexcept { return max_seq_; }
        dim_t getVocabularyLength() const noexcept { return vocab_; }
        void validate() const
        {
            if ( dim_ <= 0 || max_seq_ <= 0 || vocab_ <= 0 ) throw std::invalid_argument( "LpeConfig: embedding_dim, max_seq_len and vocab_len must be positive" );
        }
    private:
        dim_t dim_{ 0 }, max_seq_{ 0 }, vocab_{ 0 };
    };

    /// learned positional encoder: output[b, t, :] = wte[X[b, t], :] + wpe[t, :]
    template<DeviceType TDeviceType, TensorDataType TIndex, TensorDataType TPrecision>
    class Lpe : public Component<TDeviceType, TPrecision>
    {
        static_assert( TIndex == TensorDataType::INT32, "token indices are INT32" );
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using EmbeddingsTensorType = Tensor<TPrecision, MR>;
        using TokenIndexType = Tensor<TIndex, MR>;
        Lpe( const std::string& name, const LpeConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }

        EmbeddingsTensorType& forward( const TokenIndexType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Lpe must be built before calling forward()." );
            const auto& s = input.shape();
            if ( s.size() != 2 || s[ 0 ] > max_batch_ || s[ 1 ] > max_seq_ ) throw std::invalid_argument( this->getName() + ": input " + shapeToString( s ) + " exceeds the built shape" );
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            view_ = std::make_unique<EmbeddingsTensorType>( output_->view( shape_t{ s[ 0 ], s[ 1 ], config_.getEmbeddingDim() } ) );
            if constexpr ( TPrecision == TensorDataType::FP32 )
                Compute::rocmCheck( mila_cdna4_lpe_fp32( view_->data(), input.data(), wte_->data(), wpe_->data(), (int)s[ 0 ], (int)s[ 1 ], (int)config_.getEmbeddingDim(), (int)s[ 1 ],
                                                         (int)config_.getVocabularyLength(), error_flag_->data(), ctx->getStream() ) );
            else
                Compute::rocmCheck( mila_cdna4_lpe_bf16( view_->data(), input.data(), wte_->data(), wpe_->data(), (int)s[ 0 ], (int)s[ 1 ], (int)config_.getEmbeddingDim(), (int)s[ 1 ],
                                                         (int)config_.getVocabularyLength(), error_flag_->data(), ctx->getStream() ) );
            return *view_;
        }
        /// single-token step: output[b, 0, :] = wte[X[b, 0], :] + wpe[position, :] (Lpe.ixx:240-257; B rows here, the reference's view is [1, 1, C])
        EmbeddingsTensorType& decode( const TokenIndexType& input, dim_t position )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Lpe must be built before calling decode()." );
            const auto& s = input.shape();
            if ( s.size() != 2 || s[ 0 ] > max_batch_ || s[ 1 ] != 1 ) throw std::invalid_argument( this->getName() + ": decode input must be [B, 1]" );
            if ( position < 0 || position >= config_.getMaxSequenceLength() ) throw std::invalid_argument( this->getName() + ": decode position out of range" );
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            const dim_t C = config_.getEmbeddingDim();
            view_ = std::make_unique<EmbeddingsTensorType>( output_->view( shape_t{ s[ 0 ], 1, C } ) );
            if constexpr ( TPrecision == TensorDataType::FP32 )
                Compute::rocmCheck( mila_cdna4_lpe_fp32( view_->data(), input.data(), wte_->data(), wpe_->data() + static_cast<size_t>( position * C ), (int)s[ 0 ], 1, (int)C, 1,
                                                         (int)config_.getVocabularyLength(), error_flag_->data(), ctx->getStream() ) );
            else
                Compute::rocmCheck( mila_cdna4_lpe_bf16( view_->data(), input.data(), wte_->data(), wpe_->data() + static_cast<size_t>( position * C ), (int)s[ 0 ], 1, (int)C, 1,
                                                         (int)config_.getVocabularyLength(), error_flag_->data(), ctx->getStream() ) );
            return *view_;
        }
        void loadParameter( const std::string& n, const void* blob, size_t bytes ) override
        {
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            EmbeddingsTensorType* t = n == "wte" ? wte_.get() : ( n == "wpe" ? wpe_.get() : nullptr );
            if ( !t ) throw std::invalid_argument( this->getName() + ": unknown parameter '" + n + "'" );
            if ( bytes != t->sizeInBytes() ) throw std::invalid_argument( this->getName() + ": " + n + " blob size mismatch" );
            copyToDevice( *t, blob, bytes, ctx );
        }
        /// nonzero once a forward saw an id outside [0, vocab) (synchronizes the stream)
        int32_t indexError()
        {
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            int32_t v = 0;
            Compute::rocmCheck( mila_cdna4_memcpy_d2h( &v, error_flag_->data(), 4, ctx->getStream() ) );
            ctx->synchronize();
            return v;
        }
        /// parameters wte + wpe; state the [B, T, C] output and the 4-byte index-error flag
        MemoryStats getMemoryStats() const override
        {
            MemoryStats st;
            st.device_parameter_bytes = tensorBytes( wte_ ) + tensorBytes( wpe_ );
            st.device_state_bytes = tensorBytes( output_ ) + tensorBytes( error_flag_ );
            return st;
        }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 2 ) throw std::invalid_argument( this->getName() + ": build shape must be [B, T <= max_seq_len]" );
            const size_t C_ = static_cast<size_t>( config_.getEmbeddingDim() );
            MemoryStats st;
            st.device_parameter_bytes = static_cast<size_t>( config_.getVocabularyLength() + config_.getMaxSequenceLength() ) * C_ * EmbeddingsTensorType::kElemBytes;
            st.device_state_bytes = static_cast<size_t>( s[ 0 ] * s[ 1 ] ) * C_ * EmbeddingsTensorType::kElemBytes + TokenIndexType::kElemBytes;
            return st;
        }
    protected:
        /// build shape [B, T] of token ids
        void onBuilding( const BuildContext& ctx ) override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 2 || s[ 1 ] > config_.getMaxSequenceLength() ) throw std::invalid_argument( this->getName() + ": build shape must be [B, T <= max_seq_len]" );
            max_batch_ = s[ 0 ]; max_seq_ = s[ 1 ];
            const auto dev = this->getExecutionContext()->getDeviceId();
            wte_ = std::make_shared<EmbeddingsTensorType>( dev, shape_t{ config_.getVocabularyLength(), config_.getEmbeddingDim() } );
            wpe_ = std::make_shared<EmbeddingsTensorType>( dev, shape_t{ config_.getMaxSequenceLength(), config_.getEmbeddingDim() } );
            output_ = std::make_shared<EmbeddingsTensorType>( dev, shape_t{ max_batch_, max_seq_, config_.getEmbeddingDim() } );
            error_flag_ = std::make_shared<TokenIndexType>( dev, shape_t{ 1 } );
            Compute::rocmCheck( mila_cdna4_memset_zero( error_flag_->data(), 4, Compute::cast_context<TDeviceType>( this->getExecutionContext() )->getStream() ) );
        }
    private:
        LpeConfig config_;
        std::shared_ptr<EmbeddingsTensorType> wte_, wpe_, output_;
        std::unique_ptr<EmbeddingsTensorType> view_;
        std::shared_ptr<TokenIndexType> error_flag_;
        dim_t max_batch_{ 0 }, max_seq_{ 0 };
    };

    class MLPConfig
    {
    public:
        MLPConfig( dim_t input_features, dim_t hidden_size ) : in_( input_features ), hidden_( hidden_size ) {}
        template<typename Self> Self&& withBias( this Self&& s, bool b ) { s.bias_ = b; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withActivation( this Self&& s, ActivationType a ) { s.act_ = a; return std::forward<Self>( s ); }
        dim_t getInputFeatures() const noexcept { return in_; }
        dim_t getHiddenSize() const noexcept { return hidden_; }
        bool hasBias() const noexcept { return bias_; }
        ActivationType getActivationType() const noexcept { return act_; }
        void validate() const
        {
            if ( in_ <= 0 || hidden_ <= 0 ) throw std::invalid_argument( "MLPConfig: input_features and hidden_size must be positive" );
            if ( act_ != ActivationType::Gelu ) throw std::invalid_argument( "MLPConfig: the CDNA4 backend implements the Gelu MLP (GPT-2)" );
        }
    private:
        dim_t in_, hidden_;
        bool bias_{ true };
        ActivationType act_{ ActivationType::Gelu };
    };

    /// fc_1 -> gelu -> fc_2
    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class MLP : public Component<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using LinearType = Linear<TDeviceType, TPrecision>;
        using GeluType = Gelu<TDeviceType, TPrecision>;
        std::shared_ptr<LinearType> fc_1, fc_2;
        std::shared_ptr<GeluType> gelu;
        MLP( const std::string& name, const MLPConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }
        TensorType& forward( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "MLP must be built before calling forward." );
            // fc_1 -> gelu in one kernel where the backend has it (tanh GELU on bf16 GEMM rows): the same bits, without the [B, T, 4C] round trip through HBM
            if ( fc_1->fusesGelu( input.shape() ) ) return fc_2->forward( fc_1->forwardGelu( input ) );
            return fc_2->forward( gelu->forward( fc_1->forward( input ) ) );
        }
        /// single-token step (MLP.ixx:214): the same chain on a [B, 1, C] row (the Linears take their matvec branch at one row)
        TensorType& decode( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "MLP must be built before decode()." );
            return fc_2->forward( gelu->forward( fc_1->forward( input ) ) );
        }
        MemoryStats getMemoryStats() const override { MemoryStats st; if ( fc_1 ) { st += fc_1->getMemoryStats(); st += gelu->getMemoryStats(); st += fc_2->getMemoryStats(); } return st; }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            auto s = ctx.inputShape();
            MemoryStats st = fc_1->getRequiredMemory( BuildContext( s, RuntimeMode::Inference ) );
            s.back() = config_.getHiddenSize();
            st += gelu->getRequiredMemory( BuildContext( s, RuntimeMode::Inference ) );
            st += fc_2->getRequiredMemory( BuildContext( s, RuntimeMode::Inference ) );
            return st;
        }
    protected:
        void onExecutionContextSet() override
        {
            auto* ctx = this->getExecutionContext();
            fc_1 = std::make_shared<LinearType>( this->getName() + ".fc_1", LinearConfig( config_.getInputFeatures(), config_.getHiddenSize() ).withBias( config_.hasBias() ) );
            gelu = std::make_shared<GeluType>( this->getName() + ".gelu", GeluConfig() );
            fc_2 = std::make_shared<LinearType>( this->getName() + ".fc_2", LinearConfig( config_.getHiddenSize(), config_.getInputFeatures() ).withBias( config_.hasBias() ) );
            fc_1->setExecutionContext( ctx ); gelu->setExecutionContext( ctx ); fc_2->setExecutionContext( ctx );
        }
        void onBuilding( const BuildContext& ctx ) override
        {
            auto s = ctx.inputShape();
            if ( s.back() != config_.getInputFeatures() ) throw std::invalid_argument( this->getName() + ": build shape does not end in input_features" );
            fc_1->build( BuildContext( s, RuntimeMode::Inference ) );
            s.back() = config_.getHiddenSize();
            gelu->build( BuildContext( s, RuntimeMode::Inference ) );
            fc_2->build( BuildContext( s, RuntimeMode::Inference ) );
        }
    private:
        MLPConfig config_;
    };

    struct GptBlockConfig
    {
        dim_t model_dim{ 0 }, num_heads{ 0 }, hidden_size{ 0 };     ///< hidden_size 0 = 4 * model_dim (GptBlock.ixx:539-541)
        float layer_norm_eps{ 1e-5f };
        bool use_bias{ true };
        void validate() const
        {
            if ( model_dim <= 0 || num_heads <= 0 || model_dim % num_heads != 0 ) throw std::invalid_argument( "GptBlockConfig: model_dim must be a positive multiple of num_heads" );
        }
    };

    /// ln_1 -> fc_qkv_proj -> attn -> fc_out_proj -> res_1 -> ln_2 -> mlp -> res_2
    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class GptBlock : public Component<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using LayerNormType = LayerNorm<TDeviceType, TPrecision>;
        using AttentionType = MultiHeadAttention<TDeviceType, TPrecision>;
        using LinearType = Linear<TDeviceType, TPrecision>;
        using ResidualType = Residual<TDeviceType, TPrecision>;
        using MLPType = MLP<TDeviceType, TPrecision>;

        std::shared_ptr<AttentionType> attn;
        std::shared_ptr<LayerNormType> ln_1, ln_2;
        std::shared_ptr<LinearType> fc_qkv_proj, fc_out_proj;
        std::shared_ptr<ResidualType> res_1, res_2;
        std::shared_ptr<MLPType> mlp;

        GptBlock( const std::string& name, const GptBlockConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }

        TensorType& forward( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "GptBlock must be built before calling forward." );
            auto& ln1 = ln_1->forward( input );
            auto& qkv = fc_qkv_proj->forward( ln1 );
            auto& att = attn->forward( qkv );
            auto& proj = fc_out_proj->forward( att );
            auto& r1 = res_1->forward( input, proj );
            auto& ln2 = ln_2->forward( r1 );
            auto& ffn = mlp->forward( ln2 );
            return res_2->forward( r1, ffn );
        }
        /// inference-only single-token step (GptBlock.ixx:253-281): every component's forward() except attention, which is driven through decode()
        TensorType& decode( const TensorType& input, dim_t position )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "GptBlock must be built before decode()." );
            auto& ln1 = ln_1->forward( input );
            auto& qkv = fc_qkv_proj->forward( ln1 );
            auto& att = attn->decode( qkv, position );
            auto& proj = fc_out_proj->forward( att );
            auto& r1 = res_1->forward( input, proj );
            auto& ln2 = ln_2->forward( r1 );
            auto& ffn = mlp->decode( ln2 );
            return res_2->forward( r1, ffn );
        }
        bool supportsKVCache() const noexcept { return attn && attn->supportsKVCache(); }
        void initializeKVCache( dim_t max_seq_len )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "GptBlock must be built before initializeKVCache()." );
            attn->initializeKVCache( max_seq_len );
        }
        void resetKVCache() { attn->resetKVCache(); }
        /// GptBlock.ixx:364-372: the children's stats, summed
        MemoryStats getMemoryStats() const override
        {
            MemoryStats st;
            if ( !attn ) return st;
            st += attn->getMemoryStats(); st += ln_1->getMemoryStats(); st += ln_2->getMemoryStats(); st += fc_qkv_proj->getMemoryStats(); st += fc_out_proj->getMemoryStats();
            st += res_1->getMemoryStats(); st += res_2->getMemoryStats(); st += mlp->getMemoryStats();
            return st;
        }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 3 || s[ 2 ] != config_.model_dim ) throw std::invalid_argument( this->getName() + ": input must be rank 3 [B, T, model_dim]" );
            const auto inf = [&]( const shape_t& shape ) { return BuildContext( shape, RuntimeMode::Inference ); };
            MemoryStats st;
            st += ln_1->getRequiredMemory( inf( s ) ); st += ln_2->getRequiredMemory( inf( s ) ); st += fc_qkv_proj->getRequiredMemory( inf( s ) );
            st += attn->getRequiredMemory( inf( { s[ 0 ], s[ 1 ], 3 * config_.model_dim } ) ); st += fc_out_proj->getRequiredMemory( inf( s ) );
            st += res_1->getRequiredMemory( inf( s ) ); st += res_2->getRequiredMemory( inf( s ) ); st += mlp->getRequiredMemory( inf( s ) );
            return st;
        }
        std::vector<std::string> childNames() const
        {
            return { attn->getName(), ln_1->getName(), ln_2->getName(), fc_qkv_proj->getName(), fc_out_proj->getName(), res_1->getName(), res_2->getName(), mlp->getName(),
                     mlp->fc_1->getName(), mlp->gelu->getName(), mlp->fc_2->getName() };
        }
    protected:
        void onExecutionContextSet() override
        {
            const dim_t C = config_.model_dim, H = config_.hidden_size > 0 ? config_.hidden_size : 4 * C;
            auto child = [&]<typename Cmp, typename Cfg>( std::shared_ptr<Cmp>& slot, const char* leaf, const Cfg& cfg )
            {
                slot = std::make_shared<Cmp>( this->getName() + "." + leaf, cfg );
                slot->setExecutionContext( this->getExecutionContext() );
            };
            child( attn, "attn", MultiHeadAttentionConfig( C, config_.num_heads ) );
            child( ln_1, "ln_1", LayerNormConfig( shape_t{ C } ).withEpsilon( config_.layer_norm_eps ) );
            child( ln_2, "ln_2", LayerNormConfig( shape_t{ C } ).withEpsilon( config_.layer_norm_eps ) );
            child( fc_qkv_proj, "fc_qkv_proj", LinearConfig( C, 3 * C ).withBias( config_.use_bias ) );
            child( fc_out_proj, "fc_out_proj", LinearConfig( C, C ).withBias( config_.use_bias ) );
            child( res_1, "res_1", ResidualConfig{} );
            child( res_2, "res_2", ResidualConfig{} );
            child( mlp, "mlp", MLPConfig( C, H ).withBias( config_.use_bias ).withActivation( ActivationType::Gelu ) );
        }
        void onBuilding( const BuildContext& ctx ) override
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 3 || s[ 2 ] != config_.model_dim ) throw std::invalid_argument( this->getName() + ": input must be rank 3 [B, T, model_dim]" );
            const auto inf = [&]( const shape_t& shape ) { return BuildContext( shape, RuntimeMode::Inference ); };
            ln_1->build( inf( s ) ); ln_2->build( inf( s ) );
            fc_qkv_proj->build( inf( s ) );
            attn->build( inf( { s[ 0 ], s[ 1 ], 3 * config_.model_dim } ) );
            fc_out_proj->build( inf( s ) );
            res_1->build( inf( s ) ); res_2->build( inf( s ) );
            mlp->build( inf( s ) );
        }
    private:
        GptBlockConfig config_;
    };
}
