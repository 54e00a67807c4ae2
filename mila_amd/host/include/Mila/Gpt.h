// GPT-2 on the CDNA4 device (BASELINE.json configs 1-2): GptConfig and GptTransformer in the
// reference's forward order, plus its inference pair prefill() / decode() over per-block KV caches (GptTransformer.ixx:330-441, GptBlock.ixx:253-281) --
//   GptTransformer::forward  (/root/reference/Mila/Src/Dnn/Components/Transformers/Gpt/GptTransformer.ixx:221-254)
//   GptBlock::forward        (Gpt/GptBlock.ixx:148-184): ln_1 -> fc_qkv_proj -> attn -> fc_out_proj -> res_1 ->
//                            ln_2 -> MLP(fc1 -> GELU -> fc2, Components/FFN/MLP/MLP.ixx:148-161) -> res_2
//   Lpe (token + position embedding), final LayerNorm, lm_head without bias (GptTransformer.ixx:854-855).
// The reference has no BF16 rows for LayerNorm / MHA / LPE / GELU on CUDA (OPS/OperationTraits.Cuda.ixx:146-149,
// 208-222,234-236,280-282) and its CPU backend is FP32-only; the BF16 rows are added for the CDNA4 device and checked
// against the FP32 CPU oracle fed identical bf16-rounded parameters (tests/test_gpt_host_gpu.py).
#pragma once

#include <hip/hip_runtime_api.h>

#include "GptBlock.h"

namespace Mila::Dnn
{
    struct GptConfig
    {
        // GPT2_Small (Gpt.Presets.ixx, Gpt.Config.ixx:222-228)
        dim_t vocab_size = 50257, max_seq_len = 1024, embedding_dim = 768, num_layers = 12, num_heads = 12;
        float layer_norm_eps = 1e-5f;
        bool use_bias = true;
        dim_t hiddenDim() const { return 4 * embedding_dim; }
        void validate() const
        {
            if ( vocab_size <= 0 || max_seq_len <= 0 || embedding_dim <= 0 || num_layers <= 0 || num_heads <= 0 )
                throw std::invalid_argument( "GptConfig: dimensions must be positive" );
            if ( embedding_dim % num_heads != 0 ) throw std::invalid_argument( "GptConfig: embedding_dim must be a multiple of num_heads" );
            if ( embedding_dim % 32 != 0 ) throw std::invalid_argument( "GptConfig: embedding_dim must be a multiple of 32" );
        }
    };

    /// TPrecision: BF16 (BASELINE config 2) or FP32 (BASELINE config 1's model on the device: the reference's FP32 rows, OperationTraits.Cuda.ixx:50-54, :274-282)
    template<TensorDataType TPrecision>
    class GptTransformerT
    {
    public:
        static constexpr DeviceType kDevice = DeviceType::Rocm;
        static constexpr TensorDataType kPrecision = TPrecision;
        using TensorType = Tensor<kPrecision, Compute::RocmDeviceMemoryResource>;
        using TokenTensor = Tensor<TensorDataType::INT32, Compute::RocmDeviceMemoryResource>;
        using LinearType = Linear<kDevice, kPrecision>;
        using LayerNormType = LayerNorm<kDevice, kPrecision>;
        using EncoderType = Lpe<kDevice, TensorDataType::INT32, kPrecision>;
        using TransformerBlockType = GptBlock<kDevice, kPrecision>;

        /// the reference's graph (GptTransformer.ixx:828-858): lenc, tf_layer_<i>, ln_final, lm_head (no bias)
        GptTransformerT( const GptConfig& cfg, dim_t batch, dim_t seq, DeviceId device = Compute::Device::Rocm( 0 ) )
            : cfg_( cfg ), B_( batch ), T_( seq )
        {
            cfg_.validate();
            if ( batch <= 0 || seq <= 0 || seq > cfg.max_seq_len ) throw std::invalid_argument( "GptTransformer: bad batch / sequence length" );
            owned_ctx_ = Compute::createExecutionContext( device );
            ctx_ = Compute::cast_context<kDevice>( owned_ctx_.get() );
            const dim_t C = cfg_.embedding_dim;
            const BuildContext stream_ctx( shape_t{ B_, T_, C }, RuntimeMode::Inference );
            lenc_ = std::make_shared<EncoderType>( name_ + ".lenc", LpeConfig().withEmbeddingDim( C ).withMaxSequenceLength( cfg_.max_seq_len ).withVocabularyLength( cfg_.vocab_size ) );
            lenc_->setExecutionContext( ctx_ );
            lenc_->build( BuildContext( shape_t{ B_, T_ }, RuntimeMode::Inference ) );
            for ( dim_t i = 0; i < cfg_.num_layers; ++i )
            {
                GptBlockConfig bc;
                bc.model_dim = C; bc.num_heads = cfg_.num_heads; bc.hidden_size = cfg_.hiddenDim(); bc.layer_norm_eps = cfg_.layer_norm_eps; bc.use_bias = cfg_.use_bias;
                auto b = std::make_shared<TransformerBlockType>( name_ + ".tf_layer_" + std::to_string( i ), bc );
                b->setExecutionContext( ctx_ );
                b->build( stream_ctx );
                blocks_.push_back( std::move( b ) );
            }
            ln_final_ = std::make_shared<LayerNormType>( name_ + ".ln_final", LayerNormConfig( shape_t{ C } ).withEpsilon( cfg_.layer_norm_eps ) );
            ln_final_->setExecutionContext( ctx_ );
            ln_final_->build( stream_ctx );
            lm_head_ = std::make_shared<LinearType>( name_ + ".lm_head", LinearConfig( C, cfg_.vocab_size ).withBias( false ) );
            lm_head_->setExecutionContext( ctx_ );
            lm_head_->build( stream_ctx );
            ctx_->synchronize();
        }

        Compute::RocmExecutionContext* context() const noexcept { return ctx_; }
        const GptConfig& config() const noexcept { return cfg_; }

        /// GptTransformer.ixx:469-477: the children's stats (lenc, every block, ln_final, the untied lm_head) + the last-rows scratch of prefill()
        MemoryStats getMemoryStats() const
        {
            MemoryStats st = lenc_->getMemoryStats();
            for ( auto& b : blocks_ ) st += b->getMemoryStats();
            st += ln_final_->getMemoryStats();
            st += lm_head_->getMemoryStats();
            st.device_state_bytes += tensorBytes( last_rows_.get() );
            return st;
        }
        /// what a model of this configuration, batch and sequence length holds once built (KV caches come with initializeKVCache, as in the reference)
        MemoryStats getRequiredMemory() const
        {
            const BuildContext stream_ctx( shape_t{ B_, T_, cfg_.embedding_dim }, RuntimeMode::Inference );
            MemoryStats st = lenc_->getRequiredMemory( BuildContext( shape_t{ B_, T_ }, RuntimeMode::Inference ) );
            for ( auto& b : blocks_ ) st += b->getRequiredMemory( stream_ctx );
            st += ln_final_->getRequiredMemory( stream_ctx );
            st += lm_head_->getRequiredMemory( stream_ctx );
            return st;
        }

        /// parameter order of oracle/mila_oracle.c: orc_cpu_gpt2_forward (bf16 blobs on the host)
        size_t parameterCount() const { return 2 + 12 * blocks_.size() + 3; }
        void loadParameter( size_t index, const void* host_bf16, size_t bytes )
        {
            if ( index == 0 ) return lenc_->loadParameter( "wte", host_bf16, bytes );
            if ( index == 1 ) return lenc_->loadParameter( "wpe", host_bf16, bytes );
            const size_t nb = blocks_.size();
            if ( index < 2 + 12 * nb )
            {
                auto& b = *blocks_[ ( index - 2 ) / 12 ];
                switch ( ( index - 2 ) % 12 )
                {
                    case 0: return b.ln_1->loadParameter( "weight", host_bf16, bytes ); case 1: return b.ln_1->loadParameter( "bias", host_bf16, bytes );
                    case 2: return b.fc_qkv_proj->loadParameter( "weight", host_bf16, bytes ); case 3: return b.fc_qkv_proj->loadParameter( "bias", host_bf16, bytes );
                    case 4: return b.fc_out_proj->loadParameter( "weight", host_bf16, bytes ); case 5: return b.fc_out_proj->loadParameter( "bias", host_bf16, bytes );
                    case 6: return b.ln_2->loadParameter( "weight", host_bf16, bytes ); case 7: return b.ln_2->loadParameter( "bias", host_bf16, bytes );
                    case 8: return b.mlp->fc_1->loadParameter( "weight", host_bf16, bytes ); case 9: return b.mlp->fc_1->loadParameter( "bias", host_bf16, bytes );
                    case 10: return b.mlp->fc_2->loadParameter( "weight", host_bf16, bytes ); default: return b.mlp->fc_2->loadParameter( "bias", host_bf16, bytes );
                }
            }
            switch ( index - 2 - 12 * nb )
            {
                case 0: return ln_final_->loadParameter( "weight", host_bf16, bytes ); case 1: return ln_final_->loadParameter( "bias", host_bf16, bytes );
                case 2: return lm_head_->loadParameter( "weight", host_bf16, bytes );
                default: throw std::invalid_argument( "GptTransformer::loadParameter: index out of range" );
            }
        }

        /// tokens [B,T] int32 on the device -> logits [B,T,V] (bf16): lenc -> blocks -> ln_final -> lm_head (GptTransformer.ixx:221-254)
        TensorType& forward( const TokenTensor& tokens )
        {
            Compute::TraceRange tr( "gpt.forward" );
            TensorType* x = &lenc_->forward( tokens );
            for ( auto& b : blocks_ ) x = &b->forward( *x );
            return lm_head_->forward( ln_final_->forward( *x ) );
        }

        /// inference prefill (GptTransformer.ixx:330-385): the whole prompt [B, T' <= T] through lenc + blocks (each block's attention fills its KV cache), then only the
        /// last position's row through ln_final + lm_head -> logits [B, 1, V].  No chunking and no position offset: GPT-2 has learned positions
        TensorType& prefill( const TokenTensor& tokens )
        {
            Compute::TraceRange tr( "gpt.prefill" );
            const auto& s = tokens.shape();
            if ( s.size() != 2 || s[ 0 ] != B_ || s[ 1 ] <= 0 || s[ 1 ] > T_ ) throw std::invalid_argument( "GptTransformer::prefill: tokens must be [B, T <= built T]" );
            const dim_t Tp = s[ 1 ], C = cfg_.embedding_dim;
            TensorType* x = &lenc_->forward( tokens );
            for ( auto& b : blocks_ ) x = &b->forward( *x );
            if ( !last_rows_ ) last_rows_ = std::make_unique<TensorType>( ctx_->getDeviceId(), shape_t{ B_, 1, C } );
            for ( dim_t b = 0; b < B_; ++b )      // row (b, T' - 1) of every sequence (the reference's single view is this for B = 1)
                Compute::rocmCheck( mila_cdna4_memcpy_d2d( last_rows_->data() + static_cast<size_t>( b * C ), x->data() + static_cast<size_t>( ( b * Tp + Tp - 1 ) * C ),
                                                           static_cast<size_t>( C ) * TensorType::kElemBytes, ctx_->getStream() ) );
            return lm_head_->forward( ln_final_->forward( *last_rows_ ) );
        }
        /// inference-only single-token step (GptTransformer.ixx:387-441): tokens [B, 1] at absolute `position`; every block runs decode() (attention over its KV
        /// cache).  Precondition: prefill() (or forward()) has populated the caches.
        TensorType& decode( const TokenTensor& tokens, dim_t position )
        {
            Compute::TraceRange tr( "gpt.decode" );
            const auto& s = tokens.shape();
            if ( s.size() != 2 || s[ 0 ] != B_ || s[ 1 ] != 1 ) throw std::invalid_argument( "GptTransformer::decode: tokens must be [B, 1]" );
            if ( position < 0 || position >= T_ ) throw std::invalid_argument( "GptTransformer::decode: position beyond the built sequence length" );
            TensorType* x = &lenc_->decode( tokens, position );
            for ( auto& b : blocks_ ) x = &b->decode( *x, position );
            return lm_head_->forward( ln_final_->forward( *x ) );
        }
        void resetKVCache() { for ( auto& b : blocks_ ) b->resetKVCache(); }
        bool supportsKVCache() const noexcept { return !blocks_.empty() && blocks_.front()->supportsKVCache(); }

        int32_t indexError() { return lenc_->indexError(); }

        /// component names in construction order
        std::vector<std::string> componentNames() const
        {
            std::vector<std::string> out{ lenc_->getName() };
            for ( auto& b : blocks_ ) { out.push_back( b->getName() ); for ( auto& n : b->childNames() ) out.push_back( n ); }
            out.push_back( ln_final_->getName() ); out.push_back( lm_head_->getName() );
            return out;
        }

    private:
        GptConfig cfg_;
        dim_t B_, T_;
        std::string name_{ "gpt" };
        std::unique_ptr<IExecutionContext> owned_ctx_;
        Compute::RocmExecutionContext* ctx_{ nullptr };
        std::shared_ptr<EncoderType> lenc_;
        std::vector<std::shared_ptr<TransformerBlockType>> blocks_;
        std::shared_ptr<LayerNormType> ln_final_;
        std::shared_ptr<LinearType> lm_head_;
        std::unique_ptr<TensorType> last_rows_;
    };
    using GptTransformer = GptTransformerT<TensorDataType::BF16>;
    using GptTransformerFp32 = GptTransformerT<TensorDataType::FP32>;
}
