// GPT-2 on the CDNA4 device (BASELINE.json configs 1-2): GptConfig and GptTransformer in the
// reference's forward order --
//   GptTransformer::forward  (/root/reference/Mila/Src/Dnn/Components/Transformers/Gpt/GptTransformer.ixx:221-254)
//   GptBlock::forward        (Gpt/GptBlock.ixx:148-184): ln_1 -> fc_qkv_proj -> attn -> fc_out_proj -> res_1 ->
//                            ln_2 -> MLP(fc1 -> GELU -> fc2, Components/FFN/MLP/MLP.ixx:148-161) -> res_2
//   Lpe (token + position embedding), final LayerNorm, lm_head without bias (GptTransformer.ixx:854-855).
// The reference has no BF16 rows for LayerNorm / MHA / LPE / GELU on CUDA (OPS/OperationTraits.Cuda.ixx:146-149,
// 208-222,234-236,280-282) and its CPU backend is FP32-only; the BF16 rows are added for the CDNA4 device and checked
// against the FP32 CPU oracle fed identical bf16-rounded parameters (tests/test_gpt_host_gpu.py).
#pragma once

#include <hip/hip_runtime_api.h>

#include "Components.h"

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

    class GptTransformer
    {
    public:
        static constexpr DeviceType kDevice = DeviceType::Rocm;
        static constexpr TensorDataType kPrecision = TensorDataType::BF16;
        using TensorType = Tensor<kPrecision, Compute::RocmDeviceMemoryResource>;
        using TokenTensor = Tensor<TensorDataType::INT32, Compute::RocmDeviceMemoryResource>;
        using LinearType = Linear<kDevice, kPrecision>;
        using LayerNormOp = Compute::RocmLayerNormOp;

        struct Block
        {
            std::shared_ptr<TensorType> ln1_w, ln1_b, ln2_w, ln2_b;
            std::shared_ptr<LayerNormOp> ln1, ln2;
            std::shared_ptr<LinearType> qkv_proj, out_proj, fc1, fc2;
        };

        GptTransformer( const GptConfig& cfg, dim_t batch, dim_t seq, DeviceId device = Compute::Device::Rocm( 0 ) )
            : cfg_( cfg ), B_( batch ), T_( seq )
        {
            cfg_.validate();
            if ( batch <= 0 || seq <= 0 || seq > cfg.max_seq_len ) throw std::invalid_argument( "GptTransformer: bad batch / sequence length" );
            owned_ctx_ = Compute::createExecutionContext( device );
            ctx_ = Compute::cast_context<kDevice>( owned_ctx_.get() );
            const auto dev = ctx_->getDeviceId();
            const dim_t C = cfg_.embedding_dim, H = cfg_.hiddenDim();
            wte_ = std::make_shared<TensorType>( dev, shape_t{ cfg_.vocab_size, C } );
            wpe_ = std::make_shared<TensorType>( dev, shape_t{ cfg_.max_seq_len, C } );
            auto norm = [&]( std::shared_ptr<TensorType>& w, std::shared_ptr<TensorType>& b, std::shared_ptr<LayerNormOp>& op )
            {
                w = std::make_shared<TensorType>( dev, shape_t{ C } );
                b = std::make_shared<TensorType>( dev, shape_t{ C } );
                op = std::make_shared<LayerNormOp>( ctx_, Compute::NormOpConfig{ C, cfg_.layer_norm_eps, true, 0.0f } );
                op->setParameters( w.get(), b.get() );
                op->build( BuildContext( shape_t{ B_, T_, C }, RuntimeMode::Inference ) );
            };
            auto lin = [&]( const std::string& n, dim_t in, dim_t out, bool bias )
            {
                auto l = std::make_shared<LinearType>( n, LinearConfig( in, out ).withBias( bias ) );
                l->setExecutionContext( ctx_ );
                l->build( BuildContext( shape_t{ B_, T_, in }, RuntimeMode::Inference ) );
                return l;
            };
            blocks_.resize( static_cast<size_t>( cfg_.num_layers ) );
            for ( dim_t i = 0; i < cfg_.num_layers; ++i )
            {
                auto& b = blocks_[ static_cast<size_t>( i ) ];
                const std::string n = "gpt.block_" + std::to_string( i );
                norm( b.ln1_w, b.ln1_b, b.ln1 );
                norm( b.ln2_w, b.ln2_b, b.ln2 );
                b.qkv_proj = lin( n + ".fc_qkv_proj", C, 3 * C, cfg_.use_bias );
                b.out_proj = lin( n + ".fc_out_proj", C, C, cfg_.use_bias );
                b.fc1 = lin( n + ".mlp.fc1", C, H, cfg_.use_bias );
                b.fc2 = lin( n + ".mlp.fc2", H, C, cfg_.use_bias );
            }
            norm( lnf_w_, lnf_b_, lnf_ );
            lm_head_ = lin( "gpt.lm_head", C, cfg_.vocab_size, false );
            attn_ = std::make_shared<Compute::RocmMultiHeadAttentionOp>( ctx_, C, cfg_.num_heads );
            for ( auto& t : { &x_, &ln_, &att_, &res1_ } ) *t = std::make_unique<TensorType>( dev, shape_t{ B_, T_, C } );
            act_ = std::make_unique<TensorType>( dev, shape_t{ B_, T_, H } );
            err_flag_ = std::make_unique<TokenTensor>( dev, shape_t{ 1 } );
            Compute::rocmCheck( mila_cdna4_memset_zero( err_flag_->data(), 4, ctx_->getStream() ) );
            ctx_->synchronize();
        }

        Compute::RocmExecutionContext* context() const noexcept { return ctx_; }
        const GptConfig& config() const noexcept { return cfg_; }

        /// parameter order of oracle/mila_oracle.c: orc_cpu_gpt2_forward (bf16 blobs on the host)
        size_t parameterCount() const { return 2 + 12 * blocks_.size() + 3; }
        void loadParameter( size_t index, const void* host_bf16, size_t bytes )
        {
            auto up = [&]( TensorType& t ) { if ( bytes != t.sizeInBytes() ) throw std::invalid_argument( "GptTransformer::loadParameter: blob size mismatch" ); copyToDevice( t, host_bf16, bytes, ctx_ ); };
            if ( index == 0 ) return up( *wte_ );
            if ( index == 1 ) return up( *wpe_ );
            const size_t nb = blocks_.size();
            if ( index < 2 + 12 * nb )
            {
                auto& b = blocks_[ ( index - 2 ) / 12 ];
                switch ( ( index - 2 ) % 12 )
                {
                    case 0: return up( *b.ln1_w ); case 1: return up( *b.ln1_b );
                    case 2: return b.qkv_proj->loadParameter( "weight", host_bf16, bytes ); case 3: return b.qkv_proj->loadParameter( "bias", host_bf16, bytes );
                    case 4: return b.out_proj->loadParameter( "weight", host_bf16, bytes ); case 5: return b.out_proj->loadParameter( "bias", host_bf16, bytes );
                    case 6: return up( *b.ln2_w ); case 7: return up( *b.ln2_b );
                    case 8: return b.fc1->loadParameter( "weight", host_bf16, bytes ); case 9: return b.fc1->loadParameter( "bias", host_bf16, bytes );
                    case 10: return b.fc2->loadParameter( "weight", host_bf16, bytes ); default: return b.fc2->loadParameter( "bias", host_bf16, bytes );
                }
            }
            switch ( index - 2 - 12 * nb )
            {
                case 0: return up( *lnf_w_ ); case 1: return up( *lnf_b_ );
                case 2: return lm_head_->loadParameter( "weight", host_bf16, bytes );
                default: throw std::invalid_argument( "GptTransformer::loadParameter: index out of range" );
            }
        }

        /// tokens [B,T] int32 on the device -> logits [B,T,V] (bf16)
        TensorType& forward( const TokenTensor& tokens )
        {
            Compute::TraceRange tr( "gpt.forward" );
            const int B = (int)B_, T = (int)T_, C = (int)cfg_.embedding_dim;
            mila_stream_t st = ctx_->getStream();
            Compute::rocmCheck( mila_cdna4_lpe_bf16( x_->data(), tokens.data(), wte_->data(), wpe_->data(), B, T, C, T, (int)cfg_.vocab_size, err_flag_->data(), st ) );
            TensorType* x = x_.get();
            for ( auto& b : blocks_ )
            {
                b.ln1->forward( *x, *ln_ );
                auto& qkv = b.qkv_proj->forward( *ln_ );
                attn_->forward( qkv, *att_ );
                auto& proj = b.out_proj->forward( *att_ );
                Compute::rocmCheck( mila_cdna4_residual_bf16( res1_->data(), x->data(), proj.data(), (int64_t)x->size(), st ) );
                b.ln2->forward( *res1_, *ln_ );
                auto& h1 = b.fc1->forward( *ln_ );
                Compute::rocmCheck( mila_cdna4_gelu_bf16( act_->data(), h1.data(), (int64_t)h1.size(), st ) );
                auto& h2 = b.fc2->forward( *act_ );
                Compute::rocmCheck( mila_cdna4_residual_bf16( x_->data(), res1_->data(), h2.data(), (int64_t)x->size(), st ) );
                x = x_.get();
            }
            lnf_->forward( *x, *ln_ );
            return lm_head_->forward( *ln_ );
        }

        int32_t indexError()
        {
            int32_t v = 0;
            Compute::rocmCheck( mila_cdna4_memcpy_d2h( &v, err_flag_->data(), 4, ctx_->getStream() ) );
            ctx_->synchronize();
            return v;
        }

    private:
        GptConfig cfg_;
        dim_t B_, T_;
        std::unique_ptr<IExecutionContext> owned_ctx_;
        Compute::RocmExecutionContext* ctx_{ nullptr };
        std::shared_ptr<TensorType> wte_, wpe_, lnf_w_, lnf_b_;
        std::shared_ptr<LayerNormOp> lnf_;
        std::vector<Block> blocks_;
        std::shared_ptr<LinearType> lm_head_;
        std::shared_ptr<Compute::RocmMultiHeadAttentionOp> attn_;
        std::unique_ptr<TensorType> x_, ln_, att_, res1_, act_;
        std::unique_ptr<TokenTensor> err_flag_;
    };
}
