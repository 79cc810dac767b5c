"""Wire format of POST /recommend and POST /admin/corpus — field for field the reference's
src/api/schemas.py:15-70, 99-120 (the feedback models are out of scope)."""
from __future__ import annotations

from typing import Dict, List, Optional

from pydantic import BaseModel, Field, field_validator


class RecommendationRequest(BaseModel):
    query: Optional[str] = Field(default=None, description="Optional search text used as retrieval signal.")
    user_context: Optional[str] = Field(default=None, max_length=10_000,
                                        description="e.g. '[+7d w4h14] Organic Milk, Whole Wheat Bread.'")
    user_id: Optional[str] = Field(default=None, description="Resolved through eval_queries.json (demo).")
    top_k: int = Field(default=10, ge=1, le=100)
    exclude_product_ids: List[str] = Field(default_factory=list)


class RecommendationItem(BaseModel):
    product_id: str
    score: float
    product_text: Optional[str] = None


class InferenceStatistics(BaseModel):
    total_latency_ms: float
    query_embedding_time_ms: float
    similarity_compute_time_ms: float
    num_recommendations: int
    top_score: float
    avg_score: float
    timestamp: float


class RecommendationResponse(BaseModel):
    request_id: str
    recommendations: List[RecommendationItem]
    stats: Optional[InferenceStatistics] = None
    purchase_history_used: Optional[str] = None


class HealthResponse(BaseModel):
    status: str = "ok"


class CorpusUploadRequest(BaseModel):
    corpus: Dict[str, str] = Field(..., description="product_id -> product text (eval_corpus.json format)")

    @field_validator("corpus")
    @classmethod
    def corpus_non_empty(cls, v: Dict[str, str]) -> Dict[str, str]:
        if not v:
            raise ValueError("corpus must be non-empty")
        return v


class CorpusUploadResponse(BaseModel):
    status: str = "ok"
    n_products: int
