"""the kernels whose counters a round's profile keeps (tools/profile_round.sh drops every other dispatch on the GPU box; tools/summarize_profiles.py
groups by the FIRST entry that is a substring of the kernel's name, so the longer of two names that share a prefix comes first)"""
KERNELS = ("k_p_sample_chain", "k_logprob_score", "k_resnet_chain", "k_train_fused", "k_bwd_fused", "k_mlp_fwd_stash", "k_mlp_fwd", "k_se3_q_sample_target",
           "k_q_sample_target", "k_rigid_move", "k_resnet_fwd", "k_resnet_bwd", "k_resnet_dw", "k_bwd_reduce", "k_adam", "k_prep",
           "k_gemm256_bf16", "k_gemm_bf16", "k_gemm_tn256", "k_gemm_tn", "k_attn_fwd", "k_attn_bwd_dq", "k_attn_bwd_dkv", "k_ln_bf16", "k_ln_bwd_bf16",
           "k_ffn", "k_poolb", "k_embed", "k_attn")
