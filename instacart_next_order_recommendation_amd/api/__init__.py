"""HTTP surface of the reference's /recommend service, served by the MI355X recommender.

Mirrors src/api/{main,schemas,metrics,auth}.py and src/api/routes/{recommend,corpus}.py of the
reference for the hot path only; feedback storage and rate limiting are out of scope.
"""
