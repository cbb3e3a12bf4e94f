"""(package marker)"""
