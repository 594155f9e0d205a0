"""SS-MAST encoder wrapper (`src/upstream/ssmast/upstream_encoder.py:15-34` of the reference): the AST transformer with the
embedding Linear on top.  The reference builds it from modules that do not exist in its tree (`models_msn`);
the architecture is taken from `extras/mast_new/mast/models/ast_work.py` (see `src.encoder.mast`)."""
from src.encoder.mast import ASTModel


def SSMAST(config, emb_dim=256):
    pre = config["pretrain"]
    inp = pre["input"]
    be = pre["base_encoder"]
    t_dim = 1 + int(inp["length_wave"] * inp["sampling_rate"]) // 160
    if be.get("model_size", "base224") == "mvit":
        # what the reference's SS-MAST really instantiates (`extras/mast_new/mast/models_msn.py:147`): the MViTv2 encoder;
        # `base_encoder.mvit` = overrides of `mvit_engine.stage_layout` (default: configs/MVITv2_B.yaml)
        return ASTModel(label_dim=emb_dim, fstride=be.get("fstride", 10), tstride=be.get("tstride", 10), input_fdim=inp["n_mels"],
                        input_tdim=t_dim, model_size="mvit", mvit=be.get("mvit"))
    return ASTModel(label_dim=emb_dim, fstride=be.get("fstride", 10), tstride=be.get("tstride", 10), input_fdim=inp["n_mels"],
                    input_tdim=t_dim, embed_dim=be.get("output_dim", 768), depth=be.get("depth", 12),
                    num_heads=be.get("num_heads", 12))
